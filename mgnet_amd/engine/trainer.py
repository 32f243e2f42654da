"""Thin training loop reproducing what tools/train_net.py delegates to detectron2 (SURVEY 3.1 hot loop): forward ->
sum of the loss dict -> backward (bucketed all-reduce overlapped) -> full-model clip -> Adam -> poly LR.
SOLVER.AMP: bf16 activations by default (no loss scaling needed); SOLVER.AMP.DTYPE "float16" runs the reference's fp16 with
GradScaler's dynamic loss scaling evaluated on the device (solver/fused_adam.py, csrc/optim.hip: no host synchronisation)."""
import copy
import os
import time

import torch

from ..events import EventStorage
from ..solver import build_lr_scheduler, build_optimizer
from .reducer import GradReducer, ready_order


class Trainer:
    def __init__(self, cfg, model, bucket_bytes=32 << 20):
        self.cfg, self.model = cfg, model
        on_gpu = next(model.parameters()).is_cuda
        if on_gpu and cfg.SOLVER.OPTIMIZER in ("ADAM", "ADAMW", "SGD"):
            from .. import _C
            from ..solver import get_mgnet_optimizer_params
            groups = get_mgnet_optimizer_params(model, cfg.SOLVER.BASE_LR, head_lr_factor=cfg.SOLVER.HEAD_LR_FACTOR)
            plist = [p for g in groups for p in (g["params"] if isinstance(g["params"], list) else [g["params"]])]
            self.reducer = GradReducer(ready_order(model, plist), bucket_bytes, align=_C.optim_chunk(), flatten_params=True,
                                       average=False, ordered=True)
            self.optimizer = build_optimizer(cfg, model, reducer=self.reducer)
        else:
            self.optimizer = build_optimizer(cfg, model)
            self.reducer = GradReducer(ready_order(model, [p for g in self.optimizer.param_groups for p in g["params"]]), bucket_bytes,
                                       ordered=True)
        if on_gpu and self.reducer.world > 1:
            # SyncBN statistics over the peer-to-peer mailbox kernels where the node allows it (self-tested; else torch.distributed)
            from . import peer
            peer.enable()
        self.scheduler = build_lr_scheduler(cfg, self.optimizer)
        self.storage = EventStorage()
        self.iter = 0
        from ..checkpoint import Checkpointer
        self.checkpointer = Checkpointer(model, getattr(cfg, "OUTPUT_DIR", "") or "", optimizer=self.optimizer, scheduler=self.scheduler)

    def resume_or_load(self, resume=True):
        """detectron2 DefaultTrainer.resume_or_load (tools/train_net.py:234): MODEL.WEIGHTS, or the last checkpoint of
        OUTPUT_DIR (then training continues after its iteration)."""
        extra = self.checkpointer.resume_or_load(self.cfg.MODEL.WEIGHTS, resume=resume)
        if resume and self.checkpointer.has_checkpoint():
            self.iter = int(extra.get("iteration", -1)) + 1
        return extra

    def save(self, name=None):
        self._check_peers(sync=True)   # never checkpoint parameters that a timed-out statistics exchange may have poisoned
        return self.checkpointer.save(name or f"model_{self.iter - 1:07d}", iteration=self.iter - 1)

    # ---- whole-step hipGraph -----------------------------------------------------------------------------------------
    # A step issues ~950 launches from Python (about 30 ms of host time against ~38 ms of GPU time at the C4 shape); none of
    # them depends on a host value any more (OHEM branch, loss selection, bias corrections and learning rates all live in
    # device memory), so the step is captured ONCE on a capture stream and replayed: the host then only uploads the
    # learning-rate tables and calls hipGraphLaunch.
    def capture_step(self, batched_inputs):
        """Capture forward + backward + clip + Adam for `batched_inputs` (device tensors that stay alive and are refilled
        in place by the data pipeline between replays).  Needs a few eager steps before it (lazy workspaces, layout cache,
        allocator warm-up).  One process per GPU without a gradient exchange only: RCCL work inside a capture is not used."""
        assert self.reducer.world == 1, "graph replay is used by single-process runs; multi-rank steps stay eager"
        assert hasattr(self.optimizer, "launch_step"), "graph capture needs the fused optimizer"
        self.model.train()
        self._graph_inputs = batched_inputs
        import gc
        graph = torch.cuda.CUDAGraph()
        gc.collect()               # cyclic garbage of earlier steps must not be freed (allocator event queries) inside the capture
        from .. import _C
        _C.weight_cache.refresh()  # drop the rows of collected models now: the captured refresh launch keeps this table
        self._graph_keepalive = [_C.weight_cache.table]
        torch.cuda.synchronize()
        # (the captured step stays on ONE stream: capturing the side-stream branches of MGNet.forward crashes hipGraph
        #  instantiation on ROCm 7.0; the eager step with side streams is the faster of the two)
        self.model._no_side_streams = True
        try:
            with torch.cuda.graph(graph):
                self.reducer.zero_grad()
                with self.storage:
                    loss_dict = self.model(batched_inputs)
                    self._backward(loss_dict)
                self.reducer.finish()
                self.optimizer.launch_step()
        finally:   # (a failed capture must not leave the eager step without its side streams)
            self.model._no_side_streams = False
        self._graph, self._graph_losses = graph, loss_dict
        return graph

    def replay_step(self):
        """one training step = upload the per-step tables + replay the captured graph"""
        self.optimizer.prepare_step()
        self._graph.replay()
        self.scheduler.step()
        self.iter += 1
        self.storage.step()
        return self._graph_losses

    # ---- launch-plan replay (engine/plan.py) -------------------------------------------------------------------------------
    # The eager step issues ~700 launches from Python (autograd nodes, ctypes calls, torch allocations): 19-25 ms of host time against a
    # ~29 ms GPU-bound step.  One eager step is recorded as a table of launches with by-value arguments (side streams included, their
    # dependencies derived from the memory each launch touches) and replayed from C; per replay the host uploads the learning-rate
    # tables and walks the table.  Unlike the hipGraph above it keeps the concurrent branches and costs a few ms of host time per step.
    def record_plan(self, batched_inputs, prof_slots=0, best_of=1, trial_steps=6, verify_steps=0):
        """Record the step as a launch plan (see _record_plan_once).  best_of > 1 (one process only): record that many plans, run
        `trial_steps` replays of each -- every one of them a real training step -- and keep the fastest.  Why: the same step recorded twice
        in one process replays at 26.3 or at 26.7 ms, stable for the life of the recording (profiles/r05_plan_recordings.txt); what differs
        between two recordings is where the private pool's buffers landed in memory, nothing a schedule could express.  The trial timings
        are kept in `self.plan_trials`.

        verify_steps > 0 (with best_of >= 2): the two fastest recordings are replayed `verify_steps` steps each FROM THE SAME STATE
        (parameters, buffers, optimizer moments, loss scale, schedule) and must agree bit for bit in every step's losses and in the final
        gradients and parameters -- two recordings place their buffers differently and therefore run with different timing, so a
        dependency the schedule misses shows up as a difference (profiles/r05_plan_determinism.txt).  If they differ, the read-only
        declarations are switched off for this process (engine.plan.set_ro_mode("none"): every struct pointer counts as written), the step
        is recorded again and checked again; a difference that survives raises.  `self.plan_check` reports what was done."""
        if best_of <= 1 or self.reducer.world != 1:
            self.plan_trials = None
            self.plan_check = None
            return self._record_plan_once(batched_inputs, prof_slots)
        from . import plan as plan_mod
        self.plan_check = None
        for attempt in range(2):
            cands, trials = [], []
            for _ in range(best_of):
                plan = self._record_plan_once(batched_inputs, prof_slots)
                state = (self._plan, self._plan_losses, self._plan_inputs, self._plan_keep)
                for _ in range(2):
                    self.replay_plan()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(trial_steps):
                    self.replay_plan()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / trial_steps * 1e3
                trials.append(round(ms, 3))
                cands.append((ms, state))
            cands.sort(key=lambda c: c[0])
            check = None
            if verify_steps > 0:
                check = self._verify_plans(cands[0][1], cands[1][1], verify_steps)
                check["read_only_declarations"] = plan_mod.ro_mode()
            for _, st in cands[1:]:
                st[0].close()
            self._plan, self._plan_losses, self._plan_inputs, self._plan_keep = cands[0][1]
            self.plan_trials = trials
            if check is None or check["identical"]:
                if check is not None:
                    check["fallback"] = self.plan_check   # (None, or the failed check of the first attempt)
                self.plan_check = check
                return self._plan
            # two recordings of one step disagree: no schedule this process derives with the declarations honoured is trusted any more
            self.plan_check = check
            self._plan.close()
            self._plan = None
            if attempt == 1 or plan_mod.ro_mode() == "none":
                raise RuntimeError(f"two recordings of the training step replay to different results ({check}); run with --exec eager")
            plan_mod.set_ro_mode("none")
        return self._plan

    # ---- the state a step changes, for checks that run the same steps twice ------------------------------------------------------
    @torch.no_grad()
    def state_snapshot(self):
        """copies of everything a training step changes: flat parameters, module buffers (running statistics), optimizer moments / step
        count / loss scale, learning rates, schedule and iteration"""
        opt = self.optimizer
        snap = {"params": [p.detach().clone() for p in self.model.parameters()],
                "buffers": [b.detach().clone() for b in self.model.buffers()],
                "iter": self.iter, "sched": copy.deepcopy(self.scheduler.state_dict()), "lr": [g["lr"] for g in opt.param_groups]}
        if hasattr(opt, "launch_step"):
            snap["opt"] = {"t": opt._t, "m": [m.clone() for m in opt._m], "v": [None if v is None else v.clone() for v in opt._v],
                           "scaler": None if opt.scaler is None else opt.scaler.clone()}
        else:
            snap["opt_sd"] = copy.deepcopy(opt.state_dict())
        return snap

    @torch.no_grad()
    def state_restore(self, snap):
        opt = self.optimizer
        for p, q in zip(self.model.parameters(), snap["params"]):
            p.copy_(q)
        for b, q in zip(self.model.buffers(), snap["buffers"]):
            b.copy_(q)
        if "opt" in snap:
            o = snap["opt"]
            opt._t = o["t"]
            for m, q in zip(opt._m, o["m"]):
                m.copy_(q)
            for v, q in zip(opt._v, o["v"]):
                if v is not None:
                    v.copy_(q)
            if opt.scaler is not None:
                opt.scaler.copy_(o["scaler"])
            from .. import _C
            _C.weight_cache.refresh()   # the 16-bit kernel layouts are derived from the parameters at the end of every step
        else:
            opt.load_state_dict(snap["opt_sd"])
        self.scheduler.load_state_dict(copy.deepcopy(snap["sched"]))
        for g, lr in zip(opt.param_groups, snap["lr"]):
            g["lr"] = lr
        self.iter = snap["iter"]

    def _verify_plans(self, a, b, steps):
        """`steps` replays of recording `a`, then -- from the same state -- of recording `b`: every step's losses, the final gradient
        buckets and the final parameters bit for bit.  The trainer is left in the state after the steps of `b`."""
        snap = self.state_snapshot()
        out = []
        for st in (a, b):
            self.state_restore(snap)
            self._plan, self._plan_losses, self._plan_inputs, self._plan_keep = st
            losses = []
            for _ in range(steps):
                ld = self.replay_plan()
                losses.append(torch.stack([v.detach().float().reshape(()) for v in ld.values()]).clone())
            torch.cuda.synchronize()
            out.append((torch.stack(losses).cpu(), [x["flat_g"].clone() for x in self.reducer.buckets],
                        [p.detach().clone() for p in self.model.parameters()]))
        (la, ga, pa), (lb, gb, pb) = out
        same_l = la.view(torch.int32) == lb.view(torch.int32)
        first = None
        if not bool(same_l.all()):
            k = int((~same_l.all(dim=1)).nonzero()[0])
            names = list(self._plan_losses.keys())
            first = {"step": k, "losses": {names[j]: (float(la[k, j]), float(lb[k, j])) for j in range(len(names)) if not bool(same_l[k, j])}}
        grads = all(torch.equal(x, y) for x, y in zip(ga, gb))
        params = all(torch.equal(x, y) for x, y in zip(pa, pb))
        return {"what": "two recordings of the step, replayed from the same state", "steps": steps,
                "identical": first is None and grads and params, "first_difference": first, "final_gradients_equal": grads,
                "final_parameters_equal": params}

    def _record_plan_once(self, batched_inputs, prof_slots=0):
        """Record forward + backward + clip + Adam for `batched_inputs` (device tensors that stay alive and are refilled in place between
        replays) as a launch plan.  Needs a few eager steps before it (lazy workspaces, layout cache, allocator warm-up).  Single-process
        runs; raises engine.plan.PlanUnsupported when the step contains something a replay cannot express (the trainer then stays eager)."""
        from .plan import PlanUnsupported, StepPlan
        if self.reducer.world != 1:
            # multi-rank: the gradient all-reduces are host calls the plan re-issues at their place (reducer._issue); the SyncBN statistics
            # must be on the mailbox kernels (device-counted exchange numbers: replayable) -- collectives of the process group inside the
            # layers are refused by the recorder
            from . import peer
            if peer.exchange() is None:
                raise PlanUnsupported("multi-rank plan replay needs the peer-to-peer SyncBN exchange (engine/peer.py); it is not active: "
                                      + peer.report()["why"])
        if not hasattr(self.optimizer, "launch_step"):
            raise PlanUnsupported("plan replay needs the fused optimizer")
        self.model.train()
        from .. import _C
        _C.weight_cache.refresh()   # (drop the rows of collected models now: the recorded refresh launch keeps this table)
        dev = next(self.model.parameters()).device
        self.optimizer.prepare_step()

        def body():
            self.reducer.zero_grad()
            with self.storage:
                loss_dict = self.model(batched_inputs)
                self._backward(loss_dict)
            self.reducer.finish()
            self.optimizer.launch_step()
            return {k: v.detach() for k, v in loss_dict.items()}

        try:
            plan, losses = StepPlan.record(body, dev, prof_slots=prof_slots)
        except PlanUnsupported as e:
            if e.completed:   # the body ran to its end before the recorder objected: this step has been trained -- count it
                self.scheduler.step()
                self.iter += 1
                self.storage.step()
                self._check_peers()
            raise
        self.scheduler.step()
        self.iter += 1
        self.storage.step()
        self._plan, self._plan_losses, self._plan_inputs = plan, losses, batched_inputs
        self._plan_keep = [_C.weight_cache.table]
        return plan

    def replay_plan(self, prof_slot=-1):
        """one training step = upload the per-step tables + replay the recorded launches"""
        self.optimizer.prepare_step()
        self._plan.replay(prof_slot)
        self.scheduler.step()
        self.iter += 1
        self.storage.step()
        self._check_peers()
        return self._plan_losses

    # ---- the training-loop entry on top of record/replay: a NEW batch every call ------------------------------------------------
    @staticmethod
    def _batch_signature(batched_inputs):
        """what has to match for a recorded step to be replayed on a new batch: the tensors' shapes / dtypes and the frame size -- not
        `file_name`, `image_id`, ... which the dataset mapper keeps in every dict (dataset_mapper.py:129-259) and which differ per sample"""
        return tuple(tuple(sorted((k, (tuple(v.shape), v.dtype) if isinstance(v, torch.Tensor) else v)
                                  for k, v in d.items() if isinstance(v, torch.Tensor) or k in ("height", "width"))) for d in batched_inputs)

    @staticmethod
    def _layout(batched_inputs):
        """(collated buffers, how every per-frame entry views them): per-frame entries that are slices of one collated buffer
        (data/synthetic.py, data/target_generator.py) are described as (buffer index, offset, shape, stride)"""
        bases, index, layout = [], {}, []
        for d in batched_inputs:
            for k in sorted(d):
                v = d[k]
                if isinstance(v, torch.Tensor):
                    b = v._base if v._base is not None else v
                    if id(b) not in index:
                        index[id(b)] = len(bases)
                        bases.append(b)
                    layout.append((index[id(b)], v.storage_offset() - b.storage_offset(), tuple(v.shape), tuple(v.stride()), v.dtype))
        return bases, (tuple(layout), tuple((tuple(b.shape), tuple(b.stride()), b.dtype) for b in bases))

    def _static_copy(self, batched_inputs):
        """plan-owned copies of a batch with the SAME view structure, so that the recorded step keeps the zero-copy batch assembly of
        MGNet._stack and a refill is one copy per collated buffer"""
        bases, layout = self._layout(batched_inputs)
        # (host entries -- the mapper's camera_matrix -- get DEVICE copies: the recorded step must read memory a refill can rewrite)
        dev = next(self.model.parameters()).device if getattr(self, "model", None) is not None else None
        static_bases = [b.clone(memory_format=torch.preserve_format) if (dev is None or b.device == dev) else b.to(dev) for b in bases]
        index = {id(b): k for k, b in enumerate(bases)}
        out = []
        for d in batched_inputs:
            o = {}
            for k, v in d.items():
                if isinstance(v, torch.Tensor):
                    b = v._base if v._base is not None else v
                    sb = static_bases[index[id(b)]]
                    o[k] = sb.as_strided(v.shape, v.stride(), sb.storage_offset() + v.storage_offset() - b.storage_offset())
                else:
                    o[k] = v
            out.append(o)
        return out, static_bases, layout

    def _refill_static(self, batched_inputs):
        """new batch -> the static buffers the plan reads: one copy per collated buffer when the new batch is collated the same way,
        one per entry otherwise"""
        bases, layout = self._layout(batched_inputs)
        if layout == self._static_layout:
            for k, (s, sb) in enumerate(zip(bases, self._static_bases)):
                self._fill(sb, s, k)
            return
        for j, (st, d) in enumerate(zip(self._plan_inputs, batched_inputs)):
            for k, v in d.items():
                if isinstance(v, torch.Tensor):
                    self._fill(st[k], v, (j, k))

    def _fill(self, dst, src, slot):
        """dst (device, plan-owned) <- src; a host source goes through the pinned ring (a copy from pageable memory would hold the host
        until the queue has drained: the replay's 2.5 ms of host time would then be serialised with the 26 ms of the device)"""
        if src.is_cuda or not dst.is_cuda:
            dst.copy_(src, non_blocking=True)
        elif dst.is_contiguous() and (src.numel() * src.element_size()) % 4 == 0 and src.numel() > 0 and src.dtype == dst.dtype:
            from .. import _C
            if getattr(self, "_fill_stager", None) is None:
                self._fill_stager = _C.PinnedStager()
            self._fill_stager.stage_into(dst, src.contiguous(), slot=("batch", slot))
        else:
            dst.copy_(src)

    def record_plan_checked(self, batched_inputs, verify_steps=8):
        """Record the step and accept the recording only if it checks out, WITHOUT training: the state (parameters, buffers, optimizer,
        schedule, iteration) is the same after the call as before it.  Two recordings are made from that state and replayed `verify_steps`
        steps each from it; they must agree bit for bit (see record_plan).  Returns the plan (kept as the trainer's), or None with the
        reason in `self.plan_check` -- the caller then stays on the eager step.  One process only."""
        assert self.reducer.world == 1
        from . import plan as plan_mod
        snap = self.state_snapshot()
        self.plan_check = None
        for attempt in range(2):
            cands = []
            for _ in range(2):
                self.state_restore(snap)
                self._record_plan_once(batched_inputs)
                cands.append((self._plan, self._plan_losses, self._plan_inputs, self._plan_keep))
            self.state_restore(snap)
            check = self._verify_plans(cands[0], cands[1], verify_steps)
            check["read_only_declarations"] = plan_mod.ro_mode()
            check["fallback"] = self.plan_check
            self.plan_check = check
            cands[1][0].close()
            self.state_restore(snap)
            if check["identical"]:
                self._plan, self._plan_losses, self._plan_inputs, self._plan_keep = cands[0]
                return self._plan
            cands[0][0].close()
            self._plan = None
            if plan_mod.ro_mode() == "none":
                break
            plan_mod.set_ro_mode("none")
        return None

    def run_step_planned(self, batched_inputs, warmup=3, verify_steps=0):
        """`run_step` for a training loop that wants the recorded step: the first `warmup` calls run eagerly (lazy workspaces, layout cache,
        allocator), the next one records the step on plan-owned copies of the batch, every later call copies its batch into those buffers
        and replays.  A batch of another shape (or a step the recorder refuses) runs eagerly -- `self.plan_note` says why -- so the
        loop never depends on the plan.  Returns the loss dict of the step like `run_step` (device scalars, overwritten by the next replay).
        verify_steps > 0 (one process): the recording is accepted only after `record_plan_checked` -- two recordings replayed from the
        current state must agree bit for bit; the check itself trains nothing.  Refused recordings leave the loop on eager steps."""
        if getattr(self, "_plan", None) is None:
            # (eager steps THIS process has run, not `self.iter`: a run resumed from a checkpoint still has to warm up)
            if getattr(self, "_eager_steps", 0) < warmup or getattr(self, "plan_note", None):
                self._eager_steps = getattr(self, "_eager_steps", 0) + 1
                return self.run_step(batched_inputs)
            from .plan import PlanUnsupported
            static, bases, layout = self._static_copy(batched_inputs)
            err, losses = None, None
            try:
                if verify_steps > 0 and self.reducer.world == 1:
                    if self.record_plan_checked(static, verify_steps) is None:
                        raise PlanUnsupported(f"two recordings of the step replay to different results ({self.plan_check})")
                    losses = self.replay_plan()   # (the check trained nothing: this is the batch's step)
                else:
                    self.record_plan(static)
                    losses = self._plan_losses
            except PlanUnsupported as e:
                err = e
                if e.completed:        # the step ran (and was counted by record_plan) before the recorder objected: do NOT train the batch again
                    losses = e.result
            if self.reducer.world > 1:
                # every rank replays or none does: a rank on the eager path issues another launch / collective sequence than its peers expect
                import torch.distributed as dist
                ok = torch.tensor([0.0 if err is not None else 1.0], device=next(self.model.parameters()).device)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if ok.item() != 1.0 and err is None:
                    err = PlanUnsupported("the recording failed on another rank")
                    self._plan.close()
            if err is not None:
                self._plan = None
                self.plan_note = f"eager steps: {err}"
                return losses if losses is not None else self.run_step(batched_inputs)
            self._static_bases, self._static_layout, self._plan_sig = bases, layout, self._batch_signature(batched_inputs)
            return losses
        if self._batch_signature(batched_inputs) != self._plan_sig:
            self.plan_eager_steps = getattr(self, "plan_eager_steps", 0) + 1
            if self.plan_eager_steps in (10, 100, 1000):
                import logging
                logging.getLogger("mgnet_amd").warning("run_step_planned: %d steps ran eagerly because their batch does not match the recorded "
                                                       "step's tensor shapes (the plan is kept for the batches that do)", self.plan_eager_steps)
            return self.run_step(batched_inputs)
        self._refill_static(batched_inputs)
        return self.replay_plan()

    def _backward(self, loss_dict):
        """sum of the loss dict -> backward; with fp16 activations the sum is multiplied by the dynamic loss scale first
        (GradScaler.scale(losses).backward(), detectron2 AMPTrainer.run_step)"""
        scale = self.optimizer.loss_scale() if hasattr(self.optimizer, "loss_scale") else None
        # split-K sums of a bucket's convs as one launch (reducer._pack); not under hipGraph capture (the table upload uses events)
        self.reducer.lazy_wgrad(not os.environ.get("MGN_NO_LAZY_WGRAD") and not getattr(self.model, "_no_side_streams", False))
        # d(sum of the losses [* scale]) / d loss_k = 1 [* scale] for every k: the backward starts from the task losses themselves with that
        # value as their gradient instead of from a sum node -- the same gradients, but no head's backward chain begins with a tensor that
        # was computed from ALL heads' losses (on replay the heads then run forward -> loss -> backward without waiting for each other)
        vals = [v for v in loss_dict.values()]
        # (terms that carry no gradient -- a constant or detached entry of the dict -- add nothing to the sum's backward and are left out;
        #  a model whose losses are not all 0-dim takes the summed root below, as the reference's trainer does)
        live = [v for v in vals if isinstance(v, torch.Tensor) and v.requires_grad]
        if live and live[0].is_cuda and all(v.dim() == 0 for v in vals if isinstance(v, torch.Tensor)):
            g = scale.detach().reshape(()) if scale is not None else self.__dict__.get("_one")
            if g is None or g.device != live[0].device:
                g = self._one = torch.ones((), dtype=torch.float32, device=live[0].device)
            roots, grads = live, [g.to(v.dtype) if v.dtype != g.dtype else g for v in live]
        else:
            total = sum(vals)
            roots, grads = [total if scale is None else total * scale], None
        try:
            torch.autograd.backward(roots, grad_tensors=grads)
        except BaseException:
            self.reducer.abort()
            raise
        finally:
            self.reducer.lazy_wgrad(False)

    def run_step(self, batched_inputs):
        self.model.train()
        self.reducer.zero_grad()
        with self.storage:
            loss_dict = self.model(batched_inputs)
            self._backward(loss_dict)
        self.reducer.finish()
        self.optimizer.step()
        self.scheduler.step()
        self.iter += 1
        self.storage.step()
        self._check_peers()
        return loss_dict

    def _check_peers(self, sync=False):
        """a SyncBN mailbox wait that timed out is an error, not a statistic: polled EVERY step from the pinned host flag the kernel
        writes (no device synchronisation; the NaN it leaves in the statistics makes that step's losses non-finite anyway), and with a
        synchronisation before anything is written to disk"""
        if self.reducer.world > 1:
            from . import peer
            ex = peer.exchange()
            if ex is not None:
                if sync:
                    torch.cuda.synchronize()
                if ex.failed():
                    raise RuntimeError("SyncBN mailbox exchange: a peer did not post its statistics in time (a rank died or diverged in its "
                                       "launch order); the step's statistics were set to NaN. MGNET_SYNCBN=rccl uses torch.distributed "
                                       "collectives, MGNET_P2P_TIMEOUT_S sets the wait budget")
