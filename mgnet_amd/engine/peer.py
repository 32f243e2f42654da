"""Peer-to-peer exchange of the SyncBN statistics between the ranks of one node (csrc/p2p.hip): every rank pushes its row into a
mailbox in each peer's memory over xGMI and waits for the peers' flags -- one small kernel on the compute stream per exchange instead
of a torch.distributed collective (a Python -> ProcessGroupNCCL -> RCCL round trip with an event hand-shake on either side, 136 times
per step: two per InPlaceABNSync site, the reference's inplace_abn does the same through torch.distributed).

`enable(group)` (called by Trainer when the job has more than one rank) allocates the mailbox, swaps the IPC handles through the
process group once, opens the peers' mailboxes and SELF-TESTS the exchange against the process group's own all_gather on random data;
only a passing test switches ops.py over (`exchange()` returns the object), anything else -- a runtime that refuses fine-grained or IPC
memory, ranks on different hosts, a wrong or late result -- leaves the torch.distributed path in place.  MGNET_SYNCBN=rccl keeps the
collectives, =p2p makes a failing self-test an error; default "auto".

Contract with the kernels: every rank issues the same exchanges in the same order per channel; a channel is a stream (the step runs its
branches on up to three), numbered in order of first use -- which the ranks share because they run the same program."""
import ctypes
import os
import socket

import torch
import torch.distributed as dist

from .. import _C

_EX = [None]
_REPORT = {"mode": "torch.distributed", "why": "not enabled"}


def exchange():
    """the active PeerExchange, or None (then ops.py uses torch.distributed)"""
    return _EX[0]


def report():
    return dict(_REPORT)


class PeerExchange:
    def __init__(self, group, device, timeout_s=None):
        lib = _C.lib()
        # wait budget of one exchange: minutes, like the process group's own collectives (a rank may legitimately be late by a data-loader
        # warm-up, a rank-0 checkpoint or an evaluation); MGNET_P2P_TIMEOUT_S overrides.  A wait that does run out is loud (see failed()).
        if timeout_s is None:
            timeout_s = float(os.environ.get("MGNET_P2P_TIMEOUT_S", "600"))
        self.group, self.device, self.timeout_s = group, device, float(timeout_s)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if self.world > lib_const("MGN_P2P_MAX_WORLD"):
            raise RuntimeError(f"world size {self.world} > {lib_const('MGN_P2P_MAX_WORLD')}")
        # Set-up is COLLECTIVE and every rank takes the same decision at every step: a rank whose runtime refuses the mailbox still
        # takes part in the gather and in the agreement below (a rank that left early would leave its peers inside a collective)
        self._own, self._peers, handle, err = None, [], None, None
        try:
            own = ctypes.c_void_p()
            _C.check(lib.mgn_p2p_alloc(ctypes.byref(own)), "mgn_p2p_alloc")
            self._own = own.value
            handle = (ctypes.c_ubyte * 64)()
            _C.check(lib.mgn_p2p_export(self._own, handle), "mgn_p2p_export")
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        mine = (socket.gethostname(), int(torch.cuda.current_device()), None if err else bytes(handle), err)
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=group)
        refused = [f"rank {r}: {e[3]}" for r, e in enumerate(everyone) if e[2] is None]
        if refused or len({e[0] for e in everyone}) != 1:
            self._release()
            raise RuntimeError("; ".join(refused) if refused else "ranks on different hosts: the mailbox exchange is intra-node (xGMI)")
        try:
            for r, e in enumerate(everyone):
                if r == self.rank:
                    self._peers.append(self._own)
                    continue
                p = ctypes.c_void_p()
                buf = (ctypes.c_ubyte * 64).from_buffer_copy(e[2])
                _C.check(lib.mgn_p2p_open(buf, ctypes.byref(p)), "mgn_p2p_open")
                self._peers.append(p.value)
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        mapped = torch.tensor([0.0 if err else 1.0], device=device)
        dist.all_reduce(mapped, op=dist.ReduceOp.MIN, group=group)   # (also the barrier: every mailbox is mapped everywhere before anyone posts)
        if mapped.item() != 1.0:
            self._release()
            raise RuntimeError(err or "a peer could not map the mailboxes")
        self._arr = (ctypes.c_void_p * self.world)(*self._peers)
        # the time-out flag lives in pinned HOST memory the kernel writes through (system scope): the trainer polls it every step
        # without synchronising the device
        self.status = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._status_np = self.status.numpy()
        # exchange numbers per channel are counted by the kernels themselves (no per-launch host value: replayable, csrc/plan.hip)
        self.seq_dev = torch.zeros(lib_const("MGN_P2P_CHANNELS") * lib_const("MGN_P2P_MAX_WORLD"), dtype=torch.int32, device=device)
        self._chan = {}
        self.exchanges = 0

    def _release(self):
        lib = _C.lib()
        for r, p in enumerate(self._peers):
            if r != self.rank and p:
                lib.mgn_p2p_close(p)
        if self._own:
            lib.mgn_p2p_free(self._own)
        self._peers, self._own = [], None

    def _channel(self, claim=True):
        """channel of the current stream (numbered in order of first use -- the same on every rank, which runs the same program), or None
        when all channels are taken"""
        sid = torch.cuda.current_stream(self.device).cuda_stream
        c = self._chan.get(sid)
        if c is None:
            if len(self._chan) >= lib_const("MGN_P2P_CHANNELS"):
                return None
            if not claim:
                return len(self._chan)
            c = self._chan[sid] = len(self._chan)
        return c

    def release_channels(self):
        """forget which stream uses which channel (after the self-test: its side stream must not keep one of the four).  Every rank calls
        it at the same point of the program; the channels' exchange counters live on the device and simply continue."""
        self._chan = {}

    def can(self, t):
        """can this payload go through the mailbox from the current stream?  Decided identically on every rank (same shapes, same streams);
        otherwise the caller uses torch.distributed for this call."""
        return t.is_cuda and t.dtype == torch.float32 and t.numel() <= lib_const("MGN_P2P_SLOT_FLOATS") and self._channel(claim=False) is not None

    def _run(self, payload, reduce):
        payload = payload.contiguous()
        assert payload.is_cuda and payload.dtype == torch.float32 and payload.numel() <= lib_const("MGN_P2P_SLOT_FLOATS")
        n = payload.numel()
        out = torch.empty(tuple(payload.shape) if reduce else (self.world,) + tuple(payload.shape), dtype=torch.float32, device=payload.device)
        c = self._channel()
        if c is None:
            raise RuntimeError("more streams issue SyncBN exchanges than the mailbox has channels (callers check can() first)")
        _C.check(_C.lib().mgn_p2p_exchange(self._arr, self.world, self.rank, c, 0, self.seq_dev.data_ptr(), payload.data_ptr(), n, int(reduce),
                                           out.data_ptr(), self.status.data_ptr(), self.timeout_s, _C._stream()), "mgn_p2p_exchange")
        self.exchanges += 1
        return out

    def all_gather(self, t):
        """[...] fp32 -> [world, ...]: every rank's tensor, in rank order"""
        return self._run(t, False)

    def all_reduce(self, t):
        """sum over ranks in rank order (a NEW tensor; bit-identical on every rank)"""
        return self._run(t, True)

    def failed(self):
        """True if a wait ran out of time since the last call.  Reads the pinned host flag: no device synchronisation, so it sees a
        time-out of kernels that have RUN (the trainer calls it every step; before a checkpoint it synchronises first)."""
        bad = bool(self._status_np[0])
        if bad:
            self._status_np[0] = 0
        return bad

    def close(self):
        torch.cuda.synchronize(self.device)
        self._release()


_CONST = {"MGN_P2P_MAX_WORLD": 8, "MGN_P2P_CHANNELS": 4, "MGN_P2P_SLOT_FLOATS": 3072}   # include/mgnet_hip.h


def lib_const(name):
    return _CONST[name]


def _self_test(ex, rounds=96):
    """the exchange against the process group's own all_gather on random rows, on two streams (two channels), back to back without host
    synchronisation in between (so that ring slots are reused while peers lag); agreement on every rank decides.  Runs with a 2 s wait
    budget and starts with ONE synchronised exchange, so that a node on which the mailboxes do not work costs seconds, not minutes."""
    dev, group = ex.device, ex.group
    budget, ex.timeout_s = ex.timeout_s, 2.0
    try:
        probe = torch.full((8,), float(ex.rank + 1), device=dev)
        got = ex.all_reduce(probe)
        torch.cuda.synchronize(dev)
        first_ok = (not ex.failed()) and bool((got == float(ex.world * (ex.world + 1) // 2)).all())
        flag = torch.tensor([1.0 if first_ok else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if flag.item() != 1.0:
            return False
        return _self_test_burst(ex, rounds)
    finally:
        ex.timeout_s = budget
        torch.cuda.synchronize(dev)
        ex.release_channels()   # (the burst's side stream must not keep one of the four channels: training needs three)


def _self_test_burst(ex, rounds):
    dev, group = ex.device, ex.group
    g = torch.Generator(device=dev).manual_seed(1234 + ex.rank)
    # three streams = the three channels of a training step, every payload size the 68 norm sites produce (3 C forward, 2 C backward,
    # C = 64 .. 1024), a few hundred exchanges without host synchronisation so that ring slots are reused while peers lag
    sides = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    rows, got = [], []
    for k in range(rounds):
        n = (3 * 64, 2 * 64, 3 * 128, 2 * 128, 3 * 256, 2 * 256, 3 * 512, 2 * 512, 3 * 1024, 2 * 1024)[k % 10]
        t = torch.randn(n, device=dev, generator=g)
        rows.append(t)
        if k % 3:
            side = sides[k % 3 - 1]
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                got.append((ex.all_gather(t), ex.all_reduce(t)))
            torch.cuda.current_stream(dev).wait_stream(side)
        else:
            got.append((ex.all_gather(t), ex.all_reduce(t)))
    torch.cuda.synchronize(dev)
    ok = not ex.failed()
    for t, (ga, rs) in zip(rows, got):
        ref = [torch.empty_like(t) for _ in range(ex.world)]
        dist.all_gather(ref, t, group=group)
        ref = torch.stack(ref)
        want = ref[0].clone()
        for r in range(1, ex.world):
            want += ref[r]
        ok = ok and torch.equal(ga, ref) and torch.equal(rs, want)
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item() == 1.0)


def enable(group=None, device=None):
    """Switch the SyncBN statistics exchange to the mailbox kernels if the job is multi-rank on CUDA, the runtime provides fine-grained
    IPC memory and the self-test passes on every rank; returns the PeerExchange or None.  Collective: every rank must call it."""
    mode = os.environ.get("MGNET_SYNCBN", "auto").lower()
    if _EX[0] is not None:
        return _EX[0]
    if mode == "rccl" or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2 or not torch.cuda.is_available():
        _REPORT.update(mode="torch.distributed", why="MGNET_SYNCBN=rccl" if mode == "rccl" else "single rank / no CUDA")
        return None
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    ex, err = None, None
    try:
        ex = PeerExchange(group, device)
    except Exception as e:  # noqa: BLE001 -- any refusal of the runtime means: keep the collectives
        err = f"{type(e).__name__}: {e}"
    # (the decision is collective: one rank without a mailbox sends everyone back to torch.distributed)
    have = torch.tensor([0.0 if ex is None else 1.0], device=device)
    dist.all_reduce(have, op=dist.ReduceOp.MIN, group=group)
    ok = bool(have.item() == 1.0)
    if ok:
        try:
            ok = _self_test(ex)
            err = None if ok else "self-test: results differ from the process group's all_gather or a wait timed out"
        except Exception as e:  # noqa: BLE001
            ok, err = False, f"self-test raised {type(e).__name__}: {e}"
    if not ok:
        if ex is not None:
            try:
                ex.close()
            except Exception:  # noqa: BLE001
                pass
        _REPORT.update(mode="torch.distributed", why=err or "a peer could not set its mailbox up")
        if mode == "p2p":
            raise RuntimeError("MGNET_SYNCBN=p2p but the peer-to-peer exchange is not usable: " + _REPORT["why"])
        return None
    _EX[0] = ex
    _REPORT.update(mode="p2p mailbox (csrc/p2p.hip)", why="self-test passed on every rank")
    return ex


def disable():
    if _EX[0] is not None:
        try:
            _EX[0].close()
        finally:
            _EX[0] = None
            _REPORT.update(mode="torch.distributed", why="disabled")
