"""Launch-plan replay of the training step: one eager step is RECORDED -- every kernel launch of libmgnet_hip.so with its arguments by
value (csrc/mgn_launch.h, csrc/plan.hip) and the few torch ops left in the step -- and later steps are REPLAYED from C with one
`mgn_plan_run` call per segment instead of ~700 Python -> autograd -> ctypes round trips (the loop tools/train_net.py:232-234 hands
to detectron2's trainer).  hipGraph is not an option on this ROCm (hipGraphLaunch of the step costs more host time than
the eager issue and capturing the side-stream branches crashes); this keeps the side streams.

How a replay stays correct:
* static memory: the recorded step allocates from a private pool of torch's caching allocator that nothing else uses afterwards, so
  every address in the recorded arguments stays valid; the batch is a set of device tensors refilled in place; what the host changes
  per step (learning-rate tables, bias corrections) lives in device tables uploaded before the replay (FusedAdam.prepare_step);
* ordering: launches of one stream replay in their recorded order; dependencies BETWEEN streams are not copied from the host's
  wait_stream / record_stream calls (the autograd engine and the allocator add their own, and freed blocks are reused across streams
  once the HOST has seen an event complete -- none of which a replay repeats) but derived from the memory each launch reads and
  writes: pointer arguments (const = read, other = written; by-value structs are scanned for pointers) resolved to the allocator
  block they point into, a vector clock per stream, one event per needed edge.  Any two launches on different streams that touch
  overlapping memory with at least one writer are ordered as in the recording -- which also covers the reuse of a block;
* torch ops inside the step (gradient accumulations of shared feature maps, bucket packs, small host -> device table copies) are
  captured by a TorchDispatchMode with their tensors and replayed between plan segments on their stream, writing into the recorded
  output tensors (`out=` overloads); an op this cannot express aborts the recording and the trainer stays eager.
"""
import bisect
import ctypes
import gc
import os
import sys
import threading

import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from .. import _C


RO_FAMILY_CONV, RO_FAMILY_OTHER = 1, 2   # csrc/mgn_launch.h MGN_RO_FAMILY_*
_RO_MODES = ("conv", "all", "none")
_ro_override = None                       # set_ro_mode(): a process-wide override of MGN_PLAN_RO (Trainer.verify_plans falls back to "none")


def _ro_mode():
    if _ro_override is not None:
        return _ro_override
    mode = "none" if os.environ.get("MGN_PLAN_NO_RO") else os.environ.get("MGN_PLAN_RO", "conv")
    if mode not in _RO_MODES:
        raise ValueError(f"MGN_PLAN_RO={mode!r}: one of {_RO_MODES}")
    return mode


def set_ro_mode(mode):
    """which read-only declarations later recordings honour (None: back to the environment's MGN_PLAN_RO / the default `conv`)"""
    global _ro_override
    if mode is not None and mode not in _RO_MODES:
        raise ValueError(f"read-only mode {mode!r}: one of {_RO_MODES}")
    _ro_override = mode


def ro_mode():
    return _ro_mode()


class PlanUnsupported(RuntimeError):
    """the step cannot be replayed.  `completed`: the recording's body had already run to its end when the recorder's objection surfaced
    (a torch op it cannot express is only noticed, not refused) -- the step HAS been trained, `result` is what the body returned"""
    completed = False
    result = None


_ALLOC_ONLY = {"empty.memory_format", "empty_like.default", "empty_strided.default", "new_empty.default", "new_empty_strided.default"}


def _tensors(x):
    if isinstance(x, torch.Tensor):
        yield x
    elif isinstance(x, (list, tuple)):
        for y in x:
            yield from _tensors(y)


def _detached(x):
    """the same storage without autograd: a replayed torch op runs outside the graph of the recorded step"""
    if isinstance(x, torch.Tensor):
        return x.detach()
    if isinstance(x, (list, tuple)):
        return type(x)(_detached(y) for y in x)
    if isinstance(x, dict):
        return {k: _detached(v) for k, v in x.items()}
    return x


_FACTORY_OPTS = {"dtype", "layout", "device", "pin_memory", "memory_format"}


def _out_overload(func, n_out):
    """the overload of the same op that writes its result(s) into given tensors: same arguments plus `n_out` written ones
    (`out`, `grad_input`, ...) -> (overload, names of the written arguments) or (None, None)"""
    base = {a.name: str(a.type) for a in func._schema.arguments}
    order = [a.name for a in func._schema.arguments]
    pk = func.overloadpacket
    for ov_name in sorted(pk.overloads(), key=lambda n: (n != "out", n)):
        ov = getattr(pk, ov_name)
        args = ov._schema.arguments
        wr = [a.name for a in args if a.alias_info is not None and a.alias_info.is_write]
        plain = [a for a in args if not (a.alias_info is not None and a.alias_info.is_write)]
        names = [a.name for a in plain]
        dropped = [n for n in order if n not in names]   # (factory options the out= form takes from the given tensor)
        if (len(wr) == n_out and [n for n in order if n in names] == names and all(base.get(a.name) == str(a.type) for a in plain)
                and set(dropped) <= _FACTORY_OPTS and all(a.kwarg_only for a in args if a.name in wr)):
            return ov, wr
    return None, None


def _extent(t):
    """byte range of the storage a tensor lives in (conservative: the whole storage)"""
    st = t.untyped_storage()
    p = st.data_ptr()
    return (p, p + max(int(st.nbytes()), 1))


class _Blocks:
    """non-overlapping [start, end) intervals of device memory in use, newest allocation wins"""

    def __init__(self):
        self.starts, self.ends = [], []

    def add(self, s, e):
        i = bisect.bisect_left(self.starts, s)
        if i > 0 and self.ends[i - 1] > s:
            i -= 1
        j = i
        while j < len(self.starts) and self.starts[j] < e:
            j += 1
        if j - i == 1 and self.starts[i] == s and self.ends[i] == e:
            return
        self.starts[i:j] = [s]
        self.ends[i:j] = [e]

    def find(self, p):
        i = bisect.bisect_right(self.starts, p) - 1
        if i >= 0 and self.ends[i] > p:
            return (self.starts[i], self.ends[i])
        return None

    def load_snapshot(self):
        for seg in torch.cuda.memory_snapshot():
            a = seg["address"]
            for b in seg["blocks"]:
                if b["state"] != "inactive":
                    self.add(a, a + b["size"])
                a += b["size"]


class _Recorder(TorchDispatchMode):
    """collects, in issue order, the library's launches (through the proxy below) and the torch ops that do device work"""

    def __init__(self, lib, device):
        super().__init__()
        self.lib, self.device = lib, device
        self.timeline = []          # ("node", index) | ("torch", closure dict)
        self.blocks = _Blocks()
        self.blocks.load_snapshot()
        self.keep = []              # tensors / buffers the plan refers to
        # MGN_PLAN_NO_REUSE = <bytes> | all: storages up to that size stay allocated until the recording ends, so that no two tensors of
        # the step share an address (debugging aid for stale-cache-line hunts: profiles/r05_plan_determinism.txt)
        nr = os.environ.get("MGN_PLAN_NO_REUSE", "")
        self.hold_max = (1 << 62) if nr == "all" else int(nr or 0)
        self.hold = []
        self.pending_touch = ([], [])
        self.lock = threading.RLock()
        self.unresolved = set()
        self.error = None
        # memory for the small descriptor tables the step uploads once (see _C.PinnedStager.stage): allocated BEFORE the step and outside
        # its pool -- a table living in a block that an earlier launch of the step used as scratch would be overwritten on every replay
        self.arena = torch.empty(1 << 20, dtype=torch.uint8, device=device)
        self.arena_used = 0
        self.in_host_call = 0
        self._info = _C.PlanNodeInfo()
        self._offs, self._sizes, self._kinds = (ctypes.c_int * 64)(), (ctypes.c_int * 64)(), (ctypes.c_int * 64)()
        self._ro = (ctypes.c_ulonglong * 128)()
        self._fam = (ctypes.c_int * 64)()

    # ---- library calls -------------------------------------------------------------------------------------------------
    def wrap(self, name, fn):
        if name.startswith("mgn_plan_") or not callable(fn):
            return fn

        def call(*args):
            with self.lock:
                n0 = self.lib.mgn_plan_recorded()
                rc = fn(*args)
                n1 = self.lib.mgn_plan_recorded()
                if n1 > n0:
                    touch, self.pending_touch = self.pending_touch, ([], [])
                    cur = self.lib.mgn_plan_current()
                    for i in range(n0, n1):
                        st, r, w, nm = self._node_accesses(cur, i)   # (resolved NOW: the allocator's blocks change as the step goes on)
                        if i == n1 - 1:
                            r, w = r + touch[0], w + touch[1]
                        self.timeline.append(("node", dict(node=i, stream=int(st), reads=r, writes=w, name=nm)))
                return rc
        return call

    def _node_accesses(self, plan, i):
        """-> (stream, reads, writes, kernel name) of recorded node i: pointer arguments resolved to the blocks they point into"""
        lib, info, offs, sizes, kinds, blocks = self.lib, self._info, self._offs, self._sizes, self._kinds, self.blocks
        _C.check(lib.mgn_plan_node_info(plan, i, ctypes.byref(info)), "mgn_plan_node_info")
        if info.type != 0:
            return info.stream or 0, [], [], "prof"
        n = lib.mgn_plan_node_args(plan, i, len(offs), offs, sizes, kinds)
        if n < 0 or lib.mgn_plan_node_ro(plan, i, len(offs), self._ro) != n:
            raise PlanUnsupported("a kernel with more arguments than the recorder holds")
        ro = self._ro
        # Which struct arguments' read-only declarations are honoured: MGN_PLAN_RO = conv (default) | all | none, decided per FAMILY of the
        # declaration -- a constant the struct's own MGN_PLAN_RO_CONV / MGN_PLAN_RO macro fixes in csrc/ (mgn_plan_node_ro_family), not a
        # match on kernel names: a kernel joins the default set only if its argument struct is declared with MGN_PLAN_RO_CONV.
        # With ALL declarations honoured the three heads' forward passes overlap fully -- 0.3-0.7 ms faster per step -- but two replays of
        # one step then differ, in 1.5 - 8 % of the steps, by what one stale 128-pixel tile is worth (profiles/r05_plan_determinism.txt:
        # a later kernel of the SAME stream reads a tile its predecessor wrote in the previous replay; every pointer of the pair is
        # declared, plain HIP kernels do not reproduce it, any extra event nearby hides it).  Root cause not found; until it is, only the
        # convolution structs' declarations are used (0 differing steps in 3000 replays, as with none); Trainer.verify_plans() checks two
        # recordings of the real step against each other and falls back to `none` (bench.py runs it), tests/test_plan_gpu.py keeps watch.
        mode = _ro_mode()
        nm_ = info.name.decode() if info.name else ""
        fam = self._fam
        if lib.mgn_plan_node_ro_family(plan, i, len(fam), fam) != n:
            raise PlanUnsupported("a kernel with more arguments than the recorder holds")
        no_for = [x for x in os.environ.get("MGN_PLAN_NO_RO_FOR", "").split(",") if x]   # (bisecting: not for kernels named like this)
        for k in range(n):
            if mode == "none" or (mode == "conv" and fam[k] != RO_FAMILY_CONV) or any(x in nm_ for x in no_for):
                ro[2 * k] = ro[2 * k + 1] = 0
        raw = ctypes.string_at(info.blob, info.nbytes) if info.nbytes else b""
        lo = blocks.starts[0] if blocks.starts else 0
        hi = blocks.ends[-1] if blocks.ends else 0
        reads, writes = [], []
        dbg = self.__dict__.setdefault("arg_debug", {}) if os.environ.get("MGN_PLAN_DEBUG") else None
        for k in range(n):
            o, sz, kd = offs[k], sizes[k], kinds[k]
            if dbg is not None:
                if kd:
                    pp = int.from_bytes(raw[o:o + 8], "little")
                    dbg.setdefault(i, []).append((k, "const*" if kd == 1 else "ptr", pp, blocks.find(pp) if pp else None))
                elif sz >= 8:
                    for wi, pp in enumerate(np.frombuffer(raw[o:o + (sz // 8) * 8], dtype=np.uint64)):
                        if lo <= int(pp) < hi and blocks.find(int(pp)) is not None:
                            is_ro = wi < 128 and (self._ro[2 * k + wi // 64] >> (wi % 64)) & 1
                            dbg.setdefault(i, []).append((k, "struct, read-only word" if is_ro else "struct", int(pp), blocks.find(int(pp))))
            if kd:
                p = int.from_bytes(raw[o:o + 8], "little")
                if p:
                    b = blocks.find(p)
                    if b is None:
                        # library-owned memory (hipMalloc: ticket counters, workspaces): no extent known -- the address itself is the
                        # range, so that two launches on different streams holding the SAME pointer (the counter pool wraps after 256
                        # launches) are ordered as recorded; the p2p mailboxes have one channel per stream and never meet here
                        self.unresolved.add(p)
                        b = (p, p + 1)
                    (reads if kd == 1 else writes).append(b)
            elif sz >= 8:   # by-value struct: every aligned 8-byte word that points into a live block counts as read + written --
                # unless the struct declares the word a pointer its kernel only reads through (MGN_PLAN_RO in csrc/)
                words = np.frombuffer(raw[o:o + (sz // 8) * 8], dtype=np.uint64)
                for wi in np.nonzero((words >= lo) & (words < hi))[0]:
                    b = blocks.find(int(words[wi]))
                    if b is not None:
                        wi = int(wi)
                        (reads if (wi < 128 and (ro[2 * k + wi // 64] >> (wi % 64)) & 1) else writes).append(b)
        return info.stream or 0, reads, writes, info.name.decode() if info.name else "?"

    def host_call(self, fn, name, reads=(), writes=()):
        """something the HOST does inside the step that is neither a library launch nor a torch op the dispatcher shows -- a collective of
        the gradient reducer, the wait for it: executed now and again at the same place of every replay, on the stream that is current now"""
        stream = torch.cuda.current_stream(self.device)
        self.in_host_call += 1
        try:
            out = fn()
        finally:
            self.in_host_call -= 1
        with self.lock:
            self.timeline.append(("torch", dict(call=(fn, (), {}), stream=stream, name=name, reads=[_extent(t) for t in reads if t.is_cuda],
                                                writes=[_extent(t) for t in writes if t.is_cuda])))
        return out

    def static_table(self, shape, dtype):
        """a device tensor in the plan's own arena (never shared with the step's temporaries)"""
        nb = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        off = (self.arena_used + 255) // 256 * 256
        if off + nb > self.arena.numel():
            raise PlanUnsupported("the step uploads more descriptor-table bytes than the plan's arena holds")
        self.arena_used = off + nb
        return self.arena[off:off + nb].view(dtype).view(tuple(shape))

    def touch(self, reads, writes):
        """memory the NEXT library call reaches through pointers stored in device memory (descriptor tables)"""
        with self.lock:
            self.pending_touch[0].extend(_extent(t) for t in reads)
            self.pending_touch[1].extend(_extent(t) for t in writes)

    # ---- torch ops -----------------------------------------------------------------------------------------------------
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        try:
            self._note(func, args, kwargs, out)
        except PlanUnsupported as e:   # (raised later, outside the dispatcher)
            self.error = self.error or e
        return out

    def _note(self, func, args, kwargs, out):
        outs = list(_tensors(out))
        ins = list(_tensors(args)) + list(_tensors(list(kwargs.values())))
        cuda_out = [t for t in outs if t.is_cuda]
        if not cuda_out and not any(t.is_cuda for t in ins):
            return
        with self.lock:
            for t in cuda_out:
                if t.untyped_storage().nbytes():
                    self.blocks.add(*_extent(t))
                    if t.untyped_storage().nbytes() <= self.hold_max:
                        self.hold.append(t.untyped_storage())
        sch = func._schema
        name = sch.name.split("::")[-1] + "." + (func._overloadname or "default")
        if sch.name == "aten::_local_scalar_dense":
            raise PlanUnsupported("a device -> host read (.item()) inside the step")
        if self.in_host_call:
            return     # (part of a host_call: replayed as a whole)
        if sch.name.startswith("c10d"):
            raise PlanUnsupported(f"a torch.distributed collective ({name}) inside the step outside the gradient reducer -- e.g. SyncBN "
                                  "statistics on the process group instead of the peer-to-peer mailbox")
        if name in _ALLOC_ONLY or sch.name in ("aten::record_stream", "aten::is_pinned", "aten::detach",
                                                                               "aten::alias", "aten::lift_fresh"):
            return
        written = [a for a, s in zip(args, sch.arguments) if s.alias_info is not None and s.alias_info.is_write]
        written += [kwargs[s.name] for s in sch.arguments if s.name in kwargs and s.alias_info is not None and s.alias_info.is_write]
        in_ptrs = {t.untyped_storage().data_ptr() for t in ins}
        is_view = bool(outs) and not written and all(t.untyped_storage().data_ptr() in in_ptrs for t in outs)
        if is_view or (not outs and not written):
            return
        stream = torch.cuda.current_stream(self.device)
        wset = list(_tensors(written))
        args, kwargs, outs = _detached(args), _detached(kwargs), _detached(outs)
        cuda_out = [t for t in outs if t.is_cuda]
        if written:      # in-place / out= form: replays as it is
            call = (func, args, kwargs)
            w_ext = [_extent(t) for t in wset if t.is_cuda]
            r_ext = [_extent(t) for t in ins if t.is_cuda and not any(t is w for w in wset)]
        else:            # functional form: replay into the recorded outputs
            if sch.name == "aten::_to_copy" and len(outs) == 1:
                src = args[0]
                call = (torch.ops.aten.copy_.default, (outs[0], src, bool(kwargs.get("non_blocking", False))), {})
            else:
                outv, onames = _out_overload(func, len(outs))
                if outv is None:
                    raise PlanUnsupported(f"torch op {name} inside the step has no out= form")
                names = {a.name for a in outv._schema.arguments}
                kw = {k: v for k, v in kwargs.items() if k in names}
                kw.update(zip(onames, outs))
                try:
                    with torch.no_grad():
                        outv(*args, **kw)   # (validates the form now, not at the first replay; rewrites the same values)
                except Exception as e:  # noqa: BLE001
                    raise PlanUnsupported(f"torch op {name} inside the step: out= form failed ({type(e).__name__}: {e})")
                call = (outv, args, kw)
            w_ext = [_extent(t) for t in cuda_out]
            r_ext = [_extent(t) for t in ins if t.is_cuda]
        with self.lock:
            self.keep.append((args, kwargs, out))
            self.timeline.append(("torch", dict(call=call, stream=stream, name=name, reads=r_ext, writes=w_ext)))


class _LibProxy:
    def __init__(self, lib, rec):
        self._lib, self._rec, self._cache = lib, rec, {}

    def __getattr__(self, name):
        f = self._cache.get(name)
        if f is None:
            f = self._cache[name] = self._rec.wrap(name, getattr(self._lib, name))
        return f


LAUNCH, RECORD, WAIT, BREAK = 0, 1, 2, 3


def derive_schedule(items, main_id, serial=False):
    """The replay schedule of a recorded step.  `items`, in issue order: dicts with `kind` (0 library launch with `node`, 1 host closure),
    `stream` (raw handle) and `reads` / `writes` = lists of [start, end) byte ranges.  Launches of one stream keep their order; between
    streams every pair of accesses to overlapping memory with at least one writer is ordered as issued -- one vector clock per stream
    (clock[s][t] = how far into stream t stream s is known to be ordered behind), one event per edge that the clocks do not already
    imply.  Returns (ops, n_events, n_cross_stream_events, n_streams, moved): ops = [(LAUNCH, node, stream) | (RECORD, ev, stream) |
    (WAIT, ev, stream) | (BREAK, 0, stream)]; all streams are joined behind the main stream's position at the start and the main stream
    behind all of them at the end, so consecutive replays are ordered whatever the step's last launches were.  Pure host logic
    (tests/test_plan_cpu.py)."""
    streams = sorted({it["stream"] for it in items} | {main_id})
    sidx = {s: k for k, s in enumerate(streams)}
    ns = len(streams)
    clock = np.zeros((ns, ns), dtype=np.int64)        # 1-based positions, 0 = the start of the step
    pos = [0] * ns
    acc_s, acc_e, acc_stream, acc_pos, acc_w = [], [], [], [], []
    need_event = {}                                   # (stream index, pos) -> event id
    waits = [[] for _ in items]
    clock_at, where = {}, {}
    S = E = ST = PS = WR = np.zeros(0, dtype=np.int64)
    flushed = 0
    for k, it in enumerate(items):
        s = sidx[it["stream"]]
        pos[s] += 1
        mine = [(a, b, 0) for a, b in it["reads"]] + [(a, b, 1) for a, b in it["writes"]]
        if len(acc_s) > flushed:
            S = np.concatenate([S, np.array(acc_s[flushed:], dtype=np.int64)])
            E = np.concatenate([E, np.array(acc_e[flushed:], dtype=np.int64)])
            ST = np.concatenate([ST, np.array(acc_stream[flushed:], dtype=np.int64)])
            PS = np.concatenate([PS, np.array(acc_pos[flushed:], dtype=np.int64)])
            WR = np.concatenate([WR, np.array(acc_w[flushed:], dtype=np.int64)])
            flushed = len(acc_s)
        need = np.zeros(ns, dtype=np.int64)
        if serial is True or (isinstance(serial, tuple) and serial[0] <= k < serial[1]):
            # debugging aid: global program order (every launch behind every earlier one); a (lo, hi) tuple: only for items lo <= k < hi
            for t in range(ns):
                if t != s:
                    need[t] = pos[t]
        if len(S):
            for a, b, w in mine:
                m = (S < b) & (E > a) & (ST != s) & ((WR == 1) | (w == 1))
                if m.any():
                    np.maximum.at(need, ST[m], PS[m])
        for t in range(ns):
            if t != s and need[t] > clock[s][t]:
                key = (t, int(need[t]))
                ev = need_event.get(key)
                if ev is None:
                    ev = need_event[key] = len(need_event)
                waits[k].append(ev)
                clock[s] = np.maximum(clock[s], clock_at[key])
        clock[s][s] = pos[s]
        clock_at[(s, pos[s])] = clock[s].copy()
        where[(s, pos[s])] = k
        for a, b, w in mine:
            acc_s.append(a); acc_e.append(b); acc_stream.append(s); acc_pos.append(pos[s]); acc_w.append(w)
    record_after = {}
    for (t, q), ev in need_event.items():
        record_after.setdefault(where[(t, q)], []).append(ev)
    n_ev = len(need_event)
    ops = [(RECORD, n_ev, main_id)]
    ops += [(WAIT, n_ev, st) for st in streams if st != main_id]
    n_ev += 1
    moved = []
    for k, it in enumerate(items):
        for ev in waits[k]:
            ops.append((WAIT, ev, it["stream"]))
        if it["kind"] == 0:
            if it.get("moved"):
                moved.append((it["node"], it["stream"]))
            ops.append((LAUNCH, it["node"], it["stream"]))
        else:
            ops.append((BREAK, 0, it["stream"]))
        for ev in record_after.get(k, []):
            ops.append((RECORD, ev, it["stream"]))
    for st in streams:
        if st != main_id:
            ops.append((RECORD, n_ev, st))
            ops.append((WAIT, n_ev, main_id))
            n_ev += 1
    return ops, n_ev, len(need_event), ns, moved


class StepPlan:
    """`StepPlan.record(body, device)` runs `body()` (one training step: zero_grad, forward, backward, finish, optimizer launches) eagerly
    and returns the plan; `replay(prof_slot)` issues the same device work again."""

    def __init__(self):
        self.handle, self.pool, self.closures, self.keep, self.report = None, None, [], [], {}

    # ---------------------------------------------------------------------------------------------------------------------
    @classmethod
    def record(cls, body, device, prof_slots=0):
        lib = _C.lib()
        self = cls()
        dev_index = device.index if device.index is not None else torch.cuda.current_device()
        gc.collect()
        torch.cuda.synchronize(device)
        main = torch.cuda.current_stream(device)
        self.pool = torch.cuda.MemPool()
        rec = _Recorder(lib, device)
        _C.check(lib.mgn_plan_begin(), "mgn_plan_begin")
        torch._C._cuda_beginAllocateToPool(dev_index, self.pool.id)
        _C._lib = _LibProxy(lib, rec)
        _C.PLAN_RECORDER[0] = rec
        result = None
        try:
            with rec:
                result = body()
        except BaseException:
            lib.mgn_plan_abort()
            raise
        finally:
            _C._lib = lib
            _C.PLAN_RECORDER[0] = None
            torch._C._cuda_endAllocateToPool(dev_index, self.pool.id)
        if rec.error is not None:
            lib.mgn_plan_abort()
            rec.error.completed, rec.error.result = True, result
            raise rec.error
        h = ctypes.c_void_p()
        _C.check(lib.mgn_plan_end(ctypes.byref(h)), "mgn_plan_end")
        self.handle = h
        torch.cuda.synchronize(device)
        self.keep = [rec.keep, result, rec.arena]
        self._schedule(rec, main, prof_slots)
        return self, result

    # ---------------------------------------------------------------------------------------------------------------------
    def _schedule(self, rec, main, prof_slots):
        lib = _C.lib()
        self._unresolved = rec.unresolved
        items = []
        for ent in rec.timeline:
            if ent[0] == "node":
                items.append(dict(ent[1], kind=0))
            else:
                c = ent[1]
                items.append(dict(kind=1, closure=c, stream=int(c["stream"].cuda_stream), reads=c["reads"], writes=c["writes"], name=c["name"]))
        # Optional re-streaming (MGN_PLAN_SPLIT="substring[,substring]"): launches whose kernel name contains a substring move to an extra
        # stream of their own.  The dependency analysis below works from memory accesses alone, so it orders them correctly wherever they
        # run; what it buys is that their launch gaps overlap with the chain they were queued in (e.g. the weight-gradient kernels, which
        # nothing reads before the bucket is packed).
        import os
        self.extra_streams = []
        for sub in [x for x in os.environ.get("MGN_PLAN_SPLIT", "").split(",") if x]:
            st = torch.cuda.Stream(main.device)
            self.extra_streams.append(st)
            moved = 0
            for it in items:
                if it["kind"] == 0 and sub in it["name"] and "reduce" not in it["name"]:
                    it["stream"] = int(st.cuda_stream)
                    it["moved"] = True
                    moved += 1
            self.report_split = getattr(self, "report_split", {})
            self.report_split[sub] = moved
        main_id = int(main.cuda_stream)
        if os.environ.get("MGN_PLAN_MERGE"):
            # (experiment: "a>b[,c>d]" -- everything recorded on the a-th stream (handles sorted, as in MGN_PLAN_DUMP) replays on the b-th:
            #  that pair of streams is serialised in issue order, the others stay concurrent.  The dependency analysis is unchanged.)
            order = sorted({it["stream"] for it in items} | {main_id})
            tstream = {int(c["stream"].cuda_stream): c["stream"] for c in (it["closure"] for it in items if it["kind"] == 1)}
            for pair in os.environ["MGN_PLAN_MERGE"].split(","):
                a, b = (int(v) for v in pair.split(">"))
                src, dst = order[a], order[b]
                for it in items:
                    if it["stream"] == src:
                        it["stream"] = dst
                        if it["kind"] == 0:
                            it["moved"] = True
                        else:
                            it["closure"]["stream"] = tstream.get(dst) or (main if dst == main_id else torch.cuda.ExternalStream(dst, device=main.device))
                if src == main_id:
                    raise PlanUnsupported("MGN_PLAN_MERGE: the main stream cannot be merged away")
        if os.environ.get("MGN_PLAN_CLOSURE_BARRIERS"):   # (debugging: every host-issued torch op joins all streams before and after itself)
            sel = os.environ["MGN_PLAN_CLOSURE_BARRIERS"]
            for it in items:
                if it["kind"] == 1 and (sel == "1" or any(x in it["name"] for x in sel.split(","))):
                    it["writes"] = list(it["writes"]) + [(1, 2)]
                else:
                    it["reads"] = list(it["reads"]) + [(1, 2)]
        if os.environ.get("MGN_PLAN_BARRIER_AT"):   # (debugging: a full join of all streams in front of the first launch of that name)
            sub, hit, G = os.environ["MGN_PLAN_BARRIER_AT"], False, (1, 2)
            mode = os.environ.get("MGN_PLAN_BARRIER_MODE", "both")   # both | wait (X waits for everything before it) | release (everything after waits for X)
            only = os.environ.get("MGN_PLAN_BARRIER_STREAMS")       # optional: only items whose stream index (sorted handles) is listed take part
            sidx = {st: k for k, st in enumerate(sorted({it["stream"] for it in items}))}
            lo_, hi_ = int(os.environ.get("MGN_PLAN_BARRIER_FROM", 0)), int(os.environ.get("MGN_PLAN_BARRIER_TO", 1 << 30))
            for ii, it in enumerate(items):
                if not hit and it["kind"] == 0 and sub in it["name"]:
                    hit = True
                    it["writes"] = list(it["writes"]) + [G]
                    print(f"[plan barrier] at item {ii} {it['name'][:40]} stream index {sidx[it['stream']]}", file=sys.stderr)
                elif (only is None or str(sidx[it["stream"]]) in only.split(",")) and lo_ <= ii < hi_:
                    if (not hit and mode in ("both", "wait")) or (hit and mode in ("both", "release")):
                        it["reads"] = list(it["reads"]) + [G]
        serial = bool(os.environ.get("MGN_PLAN_SERIAL"))
        if os.environ.get("MGN_PLAN_SERIAL_RANGE"):   # (bisecting a missing edge: "lo:hi" = item indices that replay in program order)
            lo_s, hi_s = os.environ["MGN_PLAN_SERIAL_RANGE"].split(":")
            serial = (int(lo_s or 0), int(hi_s or len(items)))
        if os.environ.get("MGN_PLAN_DUMP"):
            with open(os.environ["MGN_PLAN_DUMP"], "a") as f:
                sid = {st: k for k, st in enumerate(sorted({it["stream"] for it in items}))}
                for k, it in enumerate(items):
                    f.write(f"{k}\t{sid[it['stream']]}\t{'K' if it['kind'] == 0 else 'T'}\t{it['name'][:90]}\n")
                f.write("----\n")
        ops, n_ev, n_cross, ns, moved_nodes = derive_schedule(items, main_id, serial=serial)
        self.closures = [it["closure"] for it in items if it["kind"] == 1]
        n = len(ops)
        types = (ctypes.c_int * n)(*[o[0] for o in ops])
        aa = (ctypes.c_int * n)(*[o[1] for o in ops])
        ss = (ctypes.c_void_p * n)(*[o[2] or None for o in ops])
        _C.check(lib.mgn_plan_compile(self.handle, n, types, aa, ss, n_ev, prof_slots), "mgn_plan_compile")
        for node, st in moved_nodes:
            _C.check(lib.mgn_plan_set_stream(self.handle, node, st), "mgn_plan_set_stream")
        self.main, self.n_ops = main, n
        # what tools/critical_path.py reads: the schedule and, per item, what it is (no tensors)
        self.ops = ops
        if os.environ.get("MGN_PLAN_DEBUG"):   # (tools/plan_why.py: which memory range orders two launches of different streams)
            self.debug_items, self.arg_debug = items, getattr(rec, "arg_debug", {})
        self.items = [dict(kind=it["kind"], node=it.get("node", -1), stream=it["stream"], name=it["name"]) for it in items]
        kernels = sum(1 for it in items if it["kind"] == 0)
        by_name = {}
        for it in items:
            if it["kind"] == 1:
                by_name[it["name"]] = by_name.get(it["name"], 0) + 1
        self.report = {"kernel_launches": kernels, "torch_ops_replayed": len(self.closures), "torch_ops": by_name, "streams": ns,
                       "cross_stream_events": n_cross, "plan_ops": n, "pointers_outside_torch_memory": len(self._unresolved)}
        if getattr(self, "report_split", None):
            self.report["moved_to_extra_streams"] = self.report_split

    # ---------------------------------------------------------------------------------------------------------------------
    def replay(self, prof_slot=-1):
        lib, h = _C.lib(), self.handle
        k = 0
        run = lib.mgn_plan_run
        for c in self.closures:
            k = run(h, k, prof_slot)
            if k < 0:
                _C.check(k, "mgn_plan_run")
            f, a, kw = c["call"]
            torch.cuda.set_stream(c["stream"])
            with torch.no_grad():
                f(*a, **kw)
        if self.closures:
            torch.cuda.set_stream(self.main)
        k = run(h, k, prof_slot)
        if k != self.n_ops:
            _C.check(k if k < 0 else -5, "mgn_plan_run")

    # ---- measuring the step from inside un-profiled replays (tools/critical_path.py) -------------------------------------------------
    def trace(self, on=True):
        """the launches of the following replays carry a hipEvent pair bound to the dispatch itself (csrc/plan.hip mgn_plan_trace)"""
        _C.check(_C.lib().mgn_plan_trace(self.handle, 1 if on else 0), "mgn_plan_trace")

    def trace_read(self):
        """after a traced replay has completed: (begin_ms, end_ms) per recorded node relative to the first launch's begin; NaN where the
        node is a mark or was skipped"""
        lib = _C.lib()
        n = lib.mgn_plan_node_count(self.handle)
        ref = next(it["node"] for it in self.items if it["kind"] == 0 and not (getattr(self, "_skipped", None) and it["node"] in self._skipped))
        a, b, d = (np.zeros(n, dtype=np.float32) for _ in range(3))
        fp = ctypes.POINTER(ctypes.c_float)
        _C.check(lib.mgn_plan_trace_read(self.handle, ref, n, a.ctypes.data_as(fp), b.ctypes.data_as(fp), d.ctypes.data_as(fp)), "mgn_plan_trace_read")
        ok = ~np.isnan(b)
        # the runtime evaluates elapsed(x, y) between dispatch-bound events as END(y) - BEGIN(x): then t_a == t_b; a runtime that
        # distinguishes the begin event would give t_b - t_a == dur
        if np.allclose(a[ok], b[ok], atol=1e-4):
            begin, end = b - d, b
        else:
            begin, end = a, b
        return begin.astype(np.float64), end.astype(np.float64)

    def set_skip(self, nodes, skip=True):
        """what-if replays: these recorded nodes are not launched (skip True / 1: results meaningless, timing = the step without them) or
        launched twice (skip 2: what they cost, as an increase, with the data intact for pure kernels); False / 0 = as recorded"""
        lib = _C.lib()
        self._skipped = getattr(self, "_skipped", set())
        mode = int(skip)
        for i in nodes:
            _C.check(lib.mgn_plan_set_skip(self.handle, int(i), mode), "mgn_plan_set_skip")
            (self._skipped.add if mode == 1 else self._skipped.discard)(int(i))

    def set_jitter(self, seed, permille=100, max_us=200):
        """race hunting: random idle kernels in front of the replayed launches (csrc/plan.hip mgn_plan_set_jitter); permille 0 = off"""
        _C.check(_C.lib().mgn_plan_set_jitter(self.handle, int(seed), int(permille), int(max_us)), "mgn_plan_set_jitter")

    def prof_elapsed_ms(self, slot):
        ms = ctypes.c_float()
        _C.check(_C.lib().mgn_plan_prof_elapsed(self.handle, slot, ctypes.byref(ms)), "mgn_plan_prof_elapsed")
        return ms.value

    def close(self):
        if self.handle is not None:
            _C.lib().mgn_plan_free(self.handle)
            self.handle = None
        self.closures, self.keep = [], []

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
