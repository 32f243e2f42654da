"""Data-parallel gradient reduction for one process per GPU over RCCL/xGMI (torch.distributed backend "nccl" on ROCm;
"gloo" in the CPU tests).  Replaces detectron2's `create_ddp_model` -> torch DDP.

* parameters are packed, in REVERSE registration order (= roughly the order in which backward produces gradients:
  heads -> decoders -> res5 .. stem; `ready_order` moves the pose network, whose backward is issued last, to the end),
  into flat fp32 buckets; `param.grad` are views into the bucket, so there is no
  gather/scatter copy around the collective.  With `flatten_params=True` the parameters themselves (and later the Adam
  moments) live in flat buffers of the same layout, which is what the fused clip+Adam kernels consume.
* a post-accumulate-grad hook counts a bucket's ready gradients and launches ONE asynchronous all-reduce per bucket as
  soon as it is complete, so the collective overlaps with the rest of backward
* xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce is per-link bound, ~2*(N-1)/N*bytes/153 GB/s
  (1.4 ms for the 123.8 MB of MGNet at N=8); 32 MB buckets keep each call far above the latency floor while leaving
  >= 4 calls to pipeline behind backward; the LAST buckets of the ready order are 16 MB and 4 MB: their all-reduce has
  nothing left to hide behind, so the exposed tail is one small call
* gradients are averaged (sum, then 1/world) like DDP -- by `finish()` or, when `average=False`, by the consumer
  (the fused optimizer folds 1/world into its clipping pass)
"""
import torch
import torch.distributed as dist


def ready_order(model, params):
    """`params` in the order in which the backward of an MGNet step hands their gradients over: reverse registration order,
    except for the pose network -- its forward is issued FIRST (lowest autograd sequence numbers, mg_net.py:262-265), so the
    engine replays its backward LAST, after the backbone's.  With plain reverse order the first bucket (log_vars + pose_net)
    completes at the very end of backward and every other bucket's all-reduce queues up behind it: no overlap.
    By default the two trunks are issued side by side (MGNet.interleaved_trunks) and their gradients arrive alternately."""
    pose = getattr(model, "pose_net", None)
    late = {id(p) for p in pose.parameters()} if pose is not None else set()
    rev = list(reversed(list(params)))
    if pose is not None and getattr(model, "interleaved_trunks", lambda: False)():
        # MGNet.forward issues pose encoder and backbone block by block beside each other (pose block k, then backbone block k), so
        # backward hands over: heads / decoders, the pose network's own convs, then backbone block k, pose block k for k = last .. stem
        wanted = {id(p) for p in rev}
        pe, bb = pose.pose_encoder, model.backbone
        units = lambda net: [net.stem] + [blk for n in net.stage_names for blk in getattr(net, n)]
        trunk = []
        for ub, up in zip(reversed(units(bb)), reversed(units(pe))):
            for u in (ub, up):
                trunk += [p for p in reversed(list(u.parameters())) if id(p) in wanted]
        in_trunk = {id(p) for p in trunk}
        assert len(in_trunk) == len(trunk)
        return [p for p in rev if id(p) not in in_trunk and id(p) not in late] + [p for p in rev if id(p) in late and id(p) not in in_trunk] + trunk
    return [p for p in rev if id(p) not in late] + [p for p in rev if id(p) in late]


class GradReducer:
    def __init__(self, params, bucket_bytes=32 << 20, group=None, align=1, flatten_params=False, average=True, ordered=False):
        """ordered: `params` is already in gradient-ready order (see `ready_order`); otherwise reverse registration order"""
        self.group, self.align, self.average = group, align, average
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        params = [p for p in params if p.requires_grad]
        self.buckets = []          # dict(flat_g, flat_p, params, offsets, pending, n)
        self._bucket_of = {}
        # Bucket sizes: 32 MB keeps each call far above the latency floor -- except at the END of the ready order: the all-reduce of the
        # last bucket cannot overlap with anything (backward is over), so the tail is made small: ... 32, 32, 16, 4 MB.  Built from the
        # end backwards, then reversed.
        seq = list(params if ordered else reversed(params))
        caps = [min(bucket_bytes, 4 << 20), min(bucket_bytes, 16 << 20)]
        groups, cur, cur_bytes = [], [], 0
        for p in reversed(seq):
            nb = self._padded(p.numel()) * 4
            cap = caps[len(groups)] if len(groups) < len(caps) else bucket_bytes
            if cur and cur_bytes + nb > cap:
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            groups.append(cur)
        for g in reversed(groups):
            self._seal(list(reversed(g)), flatten_params)
        self._handles = []
        self._next = 0             # buckets are reduced in index order on every rank (collectives must be issued in the same order)
        self.collectives = 0       # all-reduce calls issued so far (bench.py reports it)
        self.cross_stream_waits = 0  # all-reduces that had to wait for a packing copy issued on another stream
        self.enabled = True        # False: skip the exchange (bench.py's "step without all-reduce" leg)
        if self.world > 1 and flatten_params:
            # DDP broadcasts rank 0's parameters at construction; same here, on the flat buffers
            for b in self.buckets:
                dist.broadcast(b["flat_p"], src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        for p in params:
            p.register_post_accumulate_grad_hook(self._hook)

    def _padded(self, n):
        return (n + self.align - 1) // self.align * self.align

    def _seal(self, plist, flatten_params):
        offsets, off = [], 0
        for p in plist:
            assert p.dtype == torch.float32
            offsets.append(off)
            off += self._padded(p.numel())
        dev = plist[0].device
        flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        flat_p = None
        if flatten_params:
            flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
            for p, o in zip(plist, offsets):
                flat_p[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = flat_p[o:o + p.numel()].view(p.shape)
        for p, o in zip(plist, offsets):
            p.grad = flat_g[o:o + p.numel()].view(p.shape)
        b = dict(flat_g=flat_g, flat_p=flat_p, params=plist, offsets=offsets, pending=len(plist), n=len(plist), seen=set(), gstream={}, packed=True)
        self.buckets.append(b)
        for p in plist:
            self._bucket_of[p] = b

    def lazy_wgrad(self, on):
        """(de)activate deferred split-K reductions for the backward that follows (Trainer); leftovers are an error"""
        from .. import _C
        if self.buckets and self.buckets[0]["flat_g"].is_cuda:
            _C.WGRAD_LAZY[0] = id(self) if on else False

    def abort(self):
        """a backward raised: drop the deferred weight-gradient entries it registered (keyed by address, see _C.WGRAD_PENDING)"""
        from .. import _C
        _C.wgrad_pending_drop(id(self))

    def zero_grad(self):
        """Gradients are not zeroed: `.grad` is dropped, so autograd hands over each freshly computed gradient without an
        accumulate kernel per parameter; `_hook` packs a bucket's gradients into its flat buffer with one multi-tensor
        copy when the bucket is complete (parameters that got no gradient are zero-filled in `finish`)."""
        self._next = 0
        if self.buckets and self.buckets[0]["flat_g"].is_cuda:
            from .. import _C
            _C.wgrad_pending_drop(id(self))   # leftovers of a backward that raised: stale addresses must never match a later gradient
        for b in self.buckets:
            b["pending"] = b["n"]
            b["seen"] = set()
            b["gstream"] = {}
            for p in b["params"]:
                p.grad = None

    def _view(self, b, i):
        p, o = b["params"][i], b["offsets"][i]
        return b["flat_g"][o:o + p.numel()].view(p.shape)

    def _pack(self, b):
        idx = [i for i, p in enumerate(b["params"]) if p.grad is not None]
        views = [self._view(b, i) for i in idx]
        if idx and b["flat_g"].is_cuda and b["gstream"]:
            # MGNet.forward runs its independent branches on side streams and autograd replays every node (and this hook) on
            # the stream of its forward: a bucket mixes gradients produced on several streams, and the packing copy runs on the
            # stream of whichever arrived last -- it has to wait for the others and keep their memory from being recycled
            cur = torch.cuda.current_stream(b["flat_g"].device)
            for st in {s.cuda_stream: s for s in b["gstream"].values()}.values():
                if st != cur:
                    cur.wait_stream(st)
            if not torch.cuda.is_current_stream_capturing():
                for i in idx:
                    st = b["gstream"].get(id(b["params"][i]))
                    if st is not None and st != cur:
                        b["params"][i].grad.record_stream(cur)
        if idx:
            # weight gradients whose split-K reduction was deferred (_C.conv_wgrad(lazy=True)): all of the bucket's in ONE launch that
            # writes the sums straight into the bucket; everything else is copied
            from .. import _C
            lazy, plain = [], []
            for k, i in enumerate(idx):
                g = b["params"][i].grad
                ent = _C.WGRAD_PENDING.pop(g.data_ptr(), None) if g.is_cuda else None
                if ent is not None and ent[2] == tuple(g.shape):
                    lazy.append((ent[0], ent[1], views[k]))
                else:
                    plain.append(k)
            if lazy:
                if b["flat_g"].is_cuda:
                    cur = torch.cuda.current_stream(b["flat_g"].device)
                    for _d, ws, _v in lazy:
                        ws.record_stream(cur)
                _C.wgrad_reduce_batch(lazy)
            if plain:
                torch._foreach_copy_([views[k] for k in plain], [b["params"][idx[k]].grad for k in plain])
        got = set(idx)
        for i, p in enumerate(b["params"]):
            v = views[idx.index(i)] if i in got else self._view(b, i)
            if i not in got:
                v.zero_()
            p.grad = v
        b["packed"] = True
        if b["flat_g"].is_cuda and not torch.cuda.is_current_stream_capturing():
            # the all-reduce of this bucket may be issued later, from the hook of a lower-index bucket on ANOTHER stream
            # (ProcessGroupNCCL orders its stream only behind the stream that is current at the call): the event orders it
            # behind this packing copy
            ev = b.get("packed_ev")
            if ev is None:
                ev = b["packed_ev"] = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(b["flat_g"].device))
            b["packed_on"] = torch.cuda.current_stream(b["flat_g"].device)

    def _hook(self, p):
        b = self._bucket_of[p]
        if id(p) in b["seen"]:   # second accumulation into the same parameter: already counted
            return
        b["seen"].add(id(p))
        if p.is_cuda:
            b["gstream"][id(p)] = torch.cuda.current_stream(p.device)
        b["packed"] = False
        b["pending"] -= 1
        if b["pending"] == 0:
            self._pack(b)
            self._launch_ready()

    def _launch_ready(self, force=False):
        """Start the all-reduce of every complete bucket that is next in INDEX order (torch DDP's rule): a rank whose gradients
        become ready in another order -- a data-dependent branch, a loss term missing from its batch -- still issues the same
        sequence of collectives as the others.  force: the rest of the buckets (finish(): incomplete ones are packed first)."""
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            if b["pending"] > 0:
                if not force:
                    return
                self._pack(b)   # unused parameters this iteration: zero-filled, still reduced to stay in lock step
            if self.world > 1 and self.enabled:
                if b["flat_g"].is_cuda and b.get("packed_ev") is not None:
                    cur = torch.cuda.current_stream(b["flat_g"].device)
                    if b.get("packed_on") != cur:
                        cur.wait_event(b["packed_ev"])
                        self.cross_stream_waits += 1
                self._issue(lambda b=b: self._handles.append(dist.all_reduce(b["flat_g"], group=self.group, async_op=True)),
                            "grad_all_reduce", b["flat_g"])
                self.collectives += 1
            self._next += 1

    def finish(self):
        """Pack / launch the buckets whose parameters did not all get a gradient this step, wait, average."""
        self._launch_ready(force=True)
        self._next = 0
        if self.buckets and self.buckets[0]["flat_g"].is_cuda:
            from .. import _C
            # (an entry left over means autograd handed a COPY of a deferred gradient to its parameter: its values were never computed)
            assert not any(e[3] == id(self) for e in _C.WGRAD_PENDING.values()), \
                "weight gradients with a deferred split-K reduction were never packed into a bucket"
        if self.world > 1:
            self._issue(self._wait_all, "grad_all_reduce_wait", *[b["flat_g"] for b in self.buckets])
            if self.average:
                for b in self.buckets:
                    b["flat_g"].mul_(1.0 / self.world)

    def _wait_all(self):
        for h in self._handles:
            h.wait()
        self._handles.clear()

    def _issue(self, fn, name, *written):
        """a collective call / the wait for it: run now -- and, while a step is being recorded for replay (engine/plan.py), written down so
        that every replay issues it again at the same place of the launch sequence, on the same stream"""
        rec = None
        if written and written[0].is_cuda:
            from .. import _C
            rec = _C.PLAN_RECORDER[0]
        if rec is not None:
            rec.host_call(fn, name, writes=written)
        else:
            fn()

    def grad_bytes(self):
        return sum(b["flat_g"].numel() * 4 for b in self.buckets)
