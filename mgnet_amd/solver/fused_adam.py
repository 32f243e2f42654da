"""[HIP] Adam (and, round 4, AdamW / SGD) + full-model gradient-norm clipping on the flat buckets of GradReducer (mgnet_amd/csrc/optim.hip).
Same update rule as torch.optim.Adam behind FullModelGradientClippingOptimizer (tools/train_net.py:129-148); a
torch.optim.Optimizer subclass so that the LR scheduler and `param_groups` work unchanged.

Deviation from torch.optim.Adam: a parameter that received NO gradient in a step is not skipped -- GradReducer.finish zero-fills
its gradient, so its moments decay and it moves by lr * m / (sqrt(v) + eps) -- and there is one step count for the whole model.
Every registered parameter of MGNet receives a gradient in every configuration (heads that are switched off are not built), so the
trajectories agree; `load_state_dict` rejects a state whose per-parameter step counts differ."""
import numpy as np
import torch

from .. import _C


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr, reducer, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=0.0, loss_scale=None,
                 growth_interval=2000, kind="ADAM", momentum=0.9, nesterov=False):
        """kind: "ADAM" (default), "ADAMW" (torch.optim.AdamW: decoupled weight decay) or "SGD" (torch.optim.SGD with `momentum`,
        dampening 0, optional Nesterov) -- the three optimizers tools/train_net.py:129-154 builds, one fused step each"""
        assert kind in _C.OPTIM_KIND
        if kind == "SGD":
            betas = (float(momentum), 0.0)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.kind, self.nesterov = kind, bool(nesterov)
        self.reducer, self.max_grad_norm = reducer, float(max_grad_norm)
        assert reducer.align == _C.optim_chunk() and all(b["flat_p"] is not None for b in reducer.buckets)
        self.chunk = reducer.align
        dev = reducer.buckets[0]["flat_g"].device
        self._t = 0
        self._m = [torch.zeros_like(b["flat_g"]) for b in reducer.buckets]
        self._v = [torch.zeros_like(b["flat_g"]) if kind != "SGD" else None for b in reducer.buckets]   # (SGD: momentum buffer only)
        # ONE static device table for everything the host computes per step: [lr per chunk | weight decay per chunk] of every bucket,
        # then the two bias corrections -- one pinned staging copy + one device copy per step instead of three per bucket
        nch = [b["flat_g"].numel() // self.chunk for b in reducer.buckets]
        self._tab = torch.zeros(2 * sum(nch) + 2, device=dev)
        offs = np.concatenate([[0], np.cumsum(nch)])
        self._lr_dev = [self._tab[offs[k]:offs[k + 1]] for k in range(len(nch))]
        self._wd_dev = [self._tab[sum(nch) + offs[k]:sum(nch) + offs[k + 1]] for k in range(len(nch))]
        self._hyper = self._tab[2 * sum(nch):]    # [1/(1-b1^t), 1/sqrt(1-b2^t)] of the current step
        self._hyper.fill_(1.0)
        self._stager = _C.PinnedStager()
        self._partials = torch.zeros(1024 * len(reducer.buckets), device=dev)
        self._coef = torch.zeros(4, device=dev)    # clip coefficient (/ loss scale), gradient norm, found_inf
        # dynamic loss scaling (fp16 activations; torch.cuda.amp.GradScaler semantics evaluated on the device): the loss is
        # multiplied by scaler[0] before backward, mgn_clip_coef_scaled unscales / detects inf / adapts the scale and counts
        # the optimizer steps actually taken
        self.growth_interval = int(growth_interval)
        self.scaler = None if loss_scale is None else torch.tensor([float(loss_scale), 0.0, 0.0], device=dev)
        self._group_of = {p: g for g in self.param_groups for p in g["params"]}
        # chunks owned by each parameter
        self._reps = [np.array([(p.numel() + self.chunk - 1) // self.chunk for p in b["params"]]) for b in reducer.buckets]
        self._chunk_group = None

    def _upload_tables(self):
        """per-step host values -> the STATIC device tables the kernels read (lr / weight decay per chunk, bias corrections);
        staged through pinned memory, stream-ordered, no host stall.  Separate from the launches so that a captured / recorded step
        only needs this small upload before each replay."""
        g0 = self.param_groups[0]
        if self._chunk_group is None:
            # chunk -> index of its parameter's group, once: the per-step work is two gathers (the per-parameter Python loops this replaces
            # took ~0.4 ms, which the GPU spent idle at every step boundary of a replayed step)
            gidx = {id(g): k for k, g in enumerate(self.param_groups)}
            per_param = [np.array([gidx[id(self._group_of[p])] for p in b["params"]], np.int64) for b in self.reducer.buckets]
            self._chunk_group = np.concatenate([np.repeat(per_param[k], self._reps[k]) for k in range(len(per_param))])
        lr_g = np.array([g["lr"] for g in self.param_groups], np.float32)
        wd_g = np.array([g["weight_decay"] or 0.0 for g in self.param_groups], np.float32)
        hy = np.ones(2, np.float32)
        if self.scaler is None:   # (with loss scaling the device counts the steps taken: a step with inf gradients is skipped)
            bc1, bc2 = 1.0 - g0["betas"][0] ** self._t, 1.0 - g0["betas"][1] ** self._t
            hy = np.array([1.0 / bc1, 1.0 / np.sqrt(bc2)], np.float32)
        host = torch.from_numpy(np.concatenate([lr_g[self._chunk_group], wd_g[self._chunk_group], hy]))
        self._stager.stage_into(self._tab, host, slot="tables")   # (event-guarded pinned ring; the copy is a kernel on the current stream)

    def prepare_step(self):
        """host part of a step (step count, tables); `launch_step` is the device part"""
        self._t += 1
        self._upload_tables()

    @torch.no_grad()
    def launch_step(self):
        """clip + Adam + weight-layout refresh: launches only (what a captured graph contains)"""
        grad_scale = 1.0 / self.reducer.world
        n = 0
        for b in self.reducer.buckets:
            n += _C.sqnorm(b["flat_g"], self._partials, n)
        g0 = self.param_groups[0]
        if self.scaler is None:
            _C.clip_coef(self._partials, n, self.max_grad_norm, grad_scale, self._coef)
        else:
            _C.clip_coef_scaled(self._partials, n, self.max_grad_norm, grad_scale, g0["betas"][0], g0["betas"][1], self.growth_interval,
                                self.scaler, self._hyper, self._coef)
        for k, b in enumerate(self.reducer.buckets):
            _C.optim_step_dev(self.kind, b["flat_p"], b["flat_g"], self._m[k], self._v[k], self._lr_dev[k], self._wd_dev[k],
                              g0["betas"][0], g0["betas"][1], g0["eps"], self.nesterov, self._hyper, self._coef, grad_scale)
        # the kernel rewrote the flat parameter buffers behind torch's version counters: re-derive the bf16 conv layouts
        _C.weight_cache.refresh()

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        self.prepare_step()
        self.launch_step()

    # ---- torch.optim.Adam-compatible (de)serialisation: the reference's `.pth` "optimizer" entry ---------------------
    def _slices(self):
        """param -> (bucket index, offset) in the flat moment buffers"""
        return {p: (k, o) for k, b in enumerate(self.reducer.buckets) for p, o in zip(b["params"], b["offsets"])}

    def state_dict(self):
        sd = super().state_dict()   # param_groups with integer ids in group order
        where, state, idx = self._slices(), {}, 0
        steps = self._t
        if self.scaler is not None:
            # fp16: the DEVICE counts the optimizer steps actually taken (a step with inf gradients is skipped and halves the
            # scale); Adam's bias corrections continue from that count, so it is what a resume needs -- together with the scale
            # and its growth tracker (detectron2's AMPTrainer checkpoints `grad_scaler` the same way)
            sc = [float(v) for v in self.scaler.tolist()]
            sd["grad_scaler"] = {"scale": sc[0], "growth_tracker": int(sc[1]), "steps_taken": int(sc[2]),
                                 "growth_interval": self.growth_interval, "host_steps": int(self._t)}
            steps = int(sc[2])
        for g in self.param_groups:
            for p in g["params"]:
                if p not in where:   # a frozen parameter (MODEL.BACKBONE.FREEZE_AT): torch.optim keeps no state for a tensor without gradients
                    idx += 1
                    continue
                k, o = where[p]
                if self._t and self.kind == "SGD":   # torch.optim.SGD's state layout
                    state[idx] = {"momentum_buffer": self._m[k][o:o + p.numel()].view(p.shape).clone()}
                elif self._t:
                    state[idx] = {"step": torch.tensor(float(steps)),
                                  "exp_avg": self._m[k][o:o + p.numel()].view(p.shape).clone(),
                                  "exp_avg_sq": self._v[k][o:o + p.numel()].view(p.shape).clone()}
                idx += 1
        sd["state"] = state
        return sd

    @torch.no_grad()
    def load_state_dict(self, sd):
        where, idx, steps = self._slices(), 0, set()
        assert len(sd["param_groups"]) == len(self.param_groups), "optimizer state has a different number of parameter groups"
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            for key, val in sg.items():
                if key != "params":
                    g[key] = val
            for p in g["params"]:
                st = sd["state"].get(idx, sd["state"].get(str(idx)))
                if p not in where:
                    st = None
                if st is not None and self.kind == "SGD":
                    k, o = where[p]
                    if st.get("momentum_buffer") is not None:
                        self._m[k][o:o + p.numel()].copy_(st["momentum_buffer"].reshape(-1))
                elif st is not None:
                    k, o = where[p]
                    self._m[k][o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
                    self._v[k][o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                    steps.add(int(float(st["step"])))
                idx += 1
        if len(steps) > 1:
            raise ValueError(f"optimizer state with different per-parameter step counts {sorted(steps)}: torch.optim.Adam skips parameters "
                             "without a gradient and counts steps per parameter; this optimizer keeps ONE step count for the flat buffers "
                             "(every parameter of the model receives a gradient in every MGNet configuration) and cannot resume such a state")
        self._t = steps.pop() if steps else 0
        if self.scaler is not None:
            gs = sd.get("grad_scaler")
            if gs is not None:      # our own fp16 checkpoint: scale, growth tracker and the device-side step count
                self.scaler.copy_(torch.tensor([float(gs["scale"]), float(gs["growth_tracker"]), float(gs["steps_taken"])]))
                self._t = int(gs.get("host_steps", self._t))
            else:                   # a torch.optim.Adam state (reference checkpoint): continue the bias corrections from its step
                self.scaler[2] = float(self._t)

    def loss_scale(self):
        """device scalar S the loss must be multiplied with before backward (None: no loss scaling)"""
        return None if self.scaler is None else self.scaler[0]

    def grad_norm(self):
        """total gradient norm of the last step (device scalar; no sync)"""
        return self._coef[1]

    def zero_grad(self, set_to_none=False):
        self.reducer.zero_grad()
