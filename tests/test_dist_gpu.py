"""GPU, world_size 2: the data-parallel code paths with CUDA tensors.  The test box has ONE GPU, RCCL refuses two ranks on
one device, so both ranks share cuda:0 over the gloo backend: what is exercised is OUR logic (HIP SyncBN statistics with
Chan's combination, bucketed reducer, fused Adam with 1/world folded in), not the transport."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("MGNET_P2P_TIMEOUT_S", "30")   # a diverged exchange fails the test in seconds instead of hanging for the default 10 min
    import faulthandler
    faulthandler.dump_traceback_later(150, exit=True)    # a hung rank prints where it is and exits (the parent then fails on its exit code)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, fn(rank, world)))
    except BaseException as e:   # report instead of leaving the parent blocked on the queue
        import traceback
        q.put((rank, "worker failed: " + "".join(traceback.format_exception_only(type(e), e)).strip()))
        raise
    finally:
        dist.destroy_process_group()


def _spawn(fn, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get() for _ in range(world))
    for p in procs:
        p.join(180)
    failed = {r: v for r, v in out.items() if isinstance(v, str) and v.startswith("worker failed")}
    assert not failed, failed
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return out


def _syncbn_hip(rank, world):
    import torch.nn.functional as F
    from mgnet_amd.modeling import ops

    torch.manual_seed(0)
    full = (torch.randn(4, 64, 6, 10) * 2 + 1)
    w, b = torch.rand(64) + 0.5, torch.randn(64) * 0.1
    g_full = torch.randn(4, 64, 6, 10)
    sl = slice(rank * 2, (rank + 1) * 2)
    x = full[sl].cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wl, bl = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    rm, rv = torch.zeros(64, device="cuda"), torch.ones(64, device="cuda")
    y = ops.iabn(x * 1.0, wl, bl, rm, rv, True, 0.01, 1e-5, "leaky_relu", 0.01, group=dist.group.WORLD)
    (y * g_full[sl].cuda()).sum().backward()
    xf = full.double().requires_grad_(True)
    wf, bf = w.double().requires_grad_(True), b.double().requires_grad_(True)
    yf = F.leaky_relu(F.batch_norm(xf, None, None, wf.abs() + 1e-5, bf, True, 0.0, 1e-5), 0.01)
    (yf * g_full.double()).sum().backward()
    ok = torch.allclose(y.cpu().double(), yf[sl].detach(), atol=1e-4) and torch.allclose(x.grad.cpu().double(), xf.grad[sl], atol=1e-4)
    gw = wl.grad.clone()
    dist.all_reduce(gw)
    ok = ok and torch.allclose(gw.cpu().double(), wf.grad, atol=1e-3)
    return bool(ok)


def test_hip_syncbn_world2():
    assert all(_spawn(_syncbn_hip).values())


def _train(rank, world):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_network_cpu import small_model
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer

    cfg, m = small_model(with_depth=True, seed=7)   # same seed on both ranks = DDP's initial broadcast
    m = m.cuda()
    m.amp_dtype = torch.bfloat16
    tr = Trainer(cfg, m)
    assert tr.reducer.world == 2 and type(tr.optimizer).__name__ == "FusedAdam"
    batch = synthetic_batch(1, 64, 96, "cuda", seed=100 + rank)   # different data per rank
    for _ in range(3):
        out = tr.run_step(batch)
    torch.cuda.synchronize()
    finite = all(bool(torch.isfinite(v)) for v in out.values())
    chk = torch.stack([p.detach().double().sum() for p in m.parameters()]).cpu()
    other = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(other, chk)
    same = bool(torch.equal(other[0], other[1]))   # replicas stay bit-identical: same averaged gradients, same update
    bn = torch.stack([mod.running_var.double().sum() for mod in m.modules() if type(mod).__name__ == "InPlaceABNSync"]).cpu()
    obn = [torch.zeros_like(bn) for _ in range(world)]
    dist.all_gather(obn, bn)
    return finite and same and bool(torch.allclose(obn[0], obn[1], rtol=1e-6))


def test_two_rank_training_keeps_replicas_identical():
    assert all(_spawn(_train).values())


# ---- peer-to-peer SyncBN exchange (csrc/p2p.hip, engine/peer.py): two PROCESSES on the one GPU, mailboxes mapped through IPC ----------
def _p2p_exchange(rank, world):
    """mailbox all_gather / all_reduce against the process group's collectives: many rounds without host synchronisation (ring slots
    are reused while the peer lags), three streams (three channels), every payload size the 68 norm sites produce"""
    from mgnet_amd.engine import peer
    os.environ["MGNET_SYNCBN"] = "auto"
    ex = peer.enable()
    if ex is None:
        return "unavailable: " + peer.report()["why"]
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(77 + rank)
    streams = [torch.cuda.current_stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    rows, got = [], []
    for k in range(150 if world == 2 else 60):   # (more than two processes time-slice the one GPU: ~40 ms per exchange at world 8)
        C = (64, 128, 256, 512, 1024)[k % 5]
        t = torch.randn(3 if k % 2 else 2, C, device=dev, generator=g)
        rows.append(t)
        st = streams[(k // 7) % 3]
        st.wait_stream(streams[0])
        with torch.cuda.stream(st):
            if rank == 1 and k % 11 == 0:     # one rank lags: the other runs ahead into the ring
                torch.cuda._sleep(3_000_000)
            got.append((ex.all_gather(t), ex.all_reduce(t)))
        streams[0].wait_stream(st)
    torch.cuda.synchronize()
    if ex.failed():
        return "a wait timed out"
    for t, (ga, rs) in zip(rows, got):
        ref = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(ref, t)
        want = ref[0].clone()
        for r in range(1, world):
            want += ref[r]
        if not (torch.equal(ga, torch.stack(ref)) and torch.equal(rs, want)):
            return "mismatch"
    # latency of one exchange (both ranks time the same 200 exchanges)
    t = torch.randn(3, 256, device=dev)
    torch.cuda.synchronize(); dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    nlat = 200 if world == 2 else 20
    for _ in range(nlat):
        ex.all_gather(t)
    e1.record()
    torch.cuda.synchronize()
    return ("ok", round(e0.elapsed_time(e1) / nlat * 1e3, 1), peer.report()["mode"])


def test_p2p_mailbox_exchange_matches_collectives(capsys):
    out = _spawn(_p2p_exchange)
    if any(isinstance(v, str) and v.startswith("unavailable") for v in out.values()):
        pytest.skip(f"fine-grained IPC memory is not available to two processes on this box: {out}")
    assert all(isinstance(v, tuple) and v[0] == "ok" for v in out.values()), out
    with capsys.disabled():
        print(f"\n[p2p exchange] 2 processes on one GPU: {out[0][1]} / {out[1][1]} us per all_gather of 3 x 256 floats ({out[0][2]})", end="")


@pytest.mark.parametrize("world", [4, 8])
def test_p2p_mailbox_exchange_world_4_and_8(world, capsys):
    """the mailbox is laid out for 8 ranks ([channel][slot][source] indexing, 8 posting blocks while block 0 spins): 4 and 8 PROCESSES
    sharing the one GPU through IPC, same assertions as the two-rank test (150 exchanges on three channels, one rank lagging)"""
    out = _spawn(_p2p_exchange, world=world)
    if any(isinstance(v, str) and v.startswith("unavailable") for v in out.values()):
        pytest.skip(f"fine-grained IPC memory is not available to {world} processes on this box: {out}")
    assert all(isinstance(v, tuple) and v[0] == "ok" for v in out.values()), out
    with capsys.disabled():
        print(f"\n[p2p exchange] {world} processes on one GPU: {out[0][1]} us per all_gather of 3 x 256 floats", end="")


def test_p2p_mailbox_exchange_with_two_hardware_queues():
    """the three exchange streams and a spinning kernel on only TWO hardware queues (GPU_MAX_HW_QUEUES=2): every rank issues the same
    exchanges in the same order, so a wait on one queue never blocks the post it waits for"""
    os.environ["GPU_MAX_HW_QUEUES"] = "2"
    try:
        out = _spawn(_p2p_exchange, world=2)
    finally:
        del os.environ["GPU_MAX_HW_QUEUES"]
    if any(isinstance(v, str) and v.startswith("unavailable") for v in out.values()):
        pytest.skip(f"fine-grained IPC memory is not available on this box: {out}")
    assert all(isinstance(v, tuple) and v[0] == "ok" for v in out.values()), out


def _p2p_timeout(rank, world):
    """rank 1 does not post: rank 0's wait runs out of its (short) budget -> NaN in the result, host-visible flag set, no hang"""
    from mgnet_amd.engine import peer
    os.environ["MGNET_SYNCBN"] = "auto"
    ex = peer.enable()
    if ex is None:
        return "unavailable: " + peer.report()["why"]
    ex.timeout_s = 0.5
    t = torch.ones(3, 64, device="cuda")
    res = None
    if rank == 0:
        g, r = ex.all_gather(t), ex.all_reduce(t)
        torch.cuda.synchronize()
        res = (bool(torch.isnan(g).all()), bool(torch.isnan(r).all()), ex.failed(), ex.failed())
    dist.barrier()
    return res if rank == 0 else "idle"


def test_p2p_timeout_is_loud():
    out = _spawn(_p2p_timeout)
    if any(isinstance(v, str) and v.startswith("unavailable") for v in out.values()):
        pytest.skip(f"peer-to-peer exchange not available on this box: {out}")
    assert out[0] == (True, True, True, False), out   # NaN results, flag raised once and cleared by the poll


def _plan_two_ranks(rank, world):
    """launch-plan replay on two ranks: mailbox SyncBN kernels (device-counted exchange numbers) as plan nodes, the gradient all-reduces
    re-issued by the plan at their place in the launch sequence: same parameters as eager steps, bit for bit, on both ranks"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_network_cpu import small_model
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer, peer
    os.environ["MGNET_SYNCBN"] = "auto"
    res = {}
    for mode in ("eager", "plan"):
        peer.disable()
        cfg, m = small_model(with_depth=True, seed=7)
        m = m.cuda()
        m.amp_dtype = torch.bfloat16
        tr = Trainer(cfg, m)
        if peer.exchange() is None:
            return "unavailable: " + peer.report()["why"]
        batch = synthetic_batch(1, 64, 96, "cuda", seed=100 + rank)
        for _ in range(3):
            tr.run_step(batch)
        rep = None
        if mode == "plan":
            rep = tr.record_plan(batch).report
            for _ in range(2):
                tr.replay_plan()
        else:
            for _ in range(3):
                tr.run_step(batch)
        torch.cuda.synchronize()
        tr._check_peers(sync=True)
        res[mode] = (torch.stack([p.detach().double().sum() for p in m.parameters()]).cpu(), rep)
    same = bool(torch.equal(res["eager"][0], res["plan"][0]))
    other = [torch.zeros_like(res["plan"][0]) for _ in range(world)]
    dist.all_gather(other, res["plan"][0])
    return same and bool(torch.equal(other[0], other[1])), res["plan"][1]


def test_two_rank_plan_replay_equals_eager():
    out = _spawn(_plan_two_ranks)
    if any(isinstance(v, str) for v in out.values()):
        pytest.skip(f"peer-to-peer exchange not available on this box: {out}")
    assert all(v[0] is True for v in out.values()), out
    assert all(v[1]["torch_ops"].get("grad_all_reduce", 0) >= 1 for v in out.values()), out


def _p2p_refusal(rank, world):
    """one rank's runtime refuses the mailbox (alloc, then IPC open): set-up is collective, so BOTH ranks must come back with None
    (torch.distributed stays in place) -- a rank that left the set-up early would leave its peer inside a collective"""
    from mgnet_amd import _C
    from mgnet_amd.engine import peer

    real = _C.lib()
    out = []
    for entry in ("mgn_p2p_alloc", "mgn_p2p_open"):
        class Proxy:
            def __getattr__(self, name, entry=entry):
                if name == entry and rank == 1:
                    return lambda *a: -95
                return getattr(real, name)
        peer._C.lib = lambda: Proxy()
        try:
            ex = peer.enable()
        finally:
            peer._C.lib = lambda: real
        out.append((ex is None, peer.report()["mode"], peer.report()["why"]))
        if ex is not None:
            peer.disable()
    # ... and the exchange still comes up afterwards when nothing refuses
    ex = peer.enable()
    out.append(("p2p" in peer.report()["mode"]) if ex is not None else "unavailable")
    if ex is not None:
        peer.disable()
    return out


def test_p2p_setup_refused_on_one_rank_falls_back_on_both():
    out = _spawn(_p2p_refusal)
    for r in (0, 1):
        (none_a, mode_a, why_a), (none_o, mode_o, why_o), after = out[r]
        assert none_a and mode_a == "torch.distributed" and "mgn_p2p_alloc" in why_a, out
        assert none_o and mode_o == "torch.distributed" and ("mgn_p2p_open" in why_o or "could not map" in why_o), out
    assert out[0][2] == out[1][2], out


def _train_modes(rank, world):
    """the same three training steps with the statistics exchanged by the mailbox kernels and by torch.distributed: identical parameters
    (both combine the rows in rank order), and the mailbox path was really taken"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_network_cpu import small_model
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer, peer
    from mgnet_amd.modeling import ops
    res = {}
    for mode in ("auto", "rccl"):
        os.environ["MGNET_SYNCBN"] = mode
        peer.disable()
        cfg, m = small_model(with_depth=True, seed=7)
        m = m.cuda()
        m.amp_dtype = torch.bfloat16
        n0 = ops.SYNCBN_P2P[0]
        tr = Trainer(cfg, m)
        batch = synthetic_batch(1, 64, 96, "cuda", seed=100 + rank)
        for _ in range(3):
            tr.run_step(batch)
        torch.cuda.synchronize()
        res[mode] = (torch.stack([p.detach().double().sum() for p in m.parameters()]).cpu(), ops.SYNCBN_P2P[0] - n0, peer.report()["mode"])
    if res["auto"][1] == 0:
        return "unavailable: " + str(res["auto"][2])
    return bool(torch.equal(res["auto"][0], res["rccl"][0]) and res["auto"][1] >= 3 * 100 and res["rccl"][1] == 0)


def test_two_rank_training_p2p_syncbn_equals_collectives():
    out = _spawn(_train_modes)
    if any(isinstance(v, str) for v in out.values()):
        pytest.skip(f"peer-to-peer exchange not available on this box: {out}")
    assert all(v is True for v in out.values()), out


def _nccl_gather(rank, world):
    """RCCL code path of the statistics exchange (single rank: the transport is trivial, the call sequence is the real one)."""
    from mgnet_amd.modeling import ops

    stats = torch.arange(3 * 64, dtype=torch.float32, device="cuda").view(3, 64)
    g = ops._gather_stats(stats, world, dist.group.WORLD)
    sums = torch.ones(2, 64, device="cuda")
    dist.all_reduce(sums)
    torch.cuda.synchronize()
    return bool(g.shape == (world, 3, 64) and torch.equal(g[rank], stats) and float(sums.sum()) == 128.0 * world)


def _worker_nccl(rank, world, port, fn, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        q.put((rank, fn(rank, world)))
    except Exception as e:  # report instead of leaving the parent blocked
        q.put((rank, "worker failed: " + repr(e)))
    finally:
        dist.destroy_process_group()


def test_rccl_single_rank_statistics_exchange():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    p = ctx.Process(target=_worker_nccl, args=(0, 1, _free_port(), _nccl_gather, q))
    p.start()
    rank, ok = q.get()
    p.join(120)
    assert ok is True, ok


@pytest.mark.parametrize("gpus,how", [(2, "auto"), (2, "plan"), (8, "auto")])
def test_bench_self_launch_end_to_end_on_gloo(gpus, how):
    """`python bench.py --gpus N` as the driver runs it (self-launch of the N ranks, barrier + max-over-ranks timing, one JSON line with
    config.distributed) -- here with MGNET_DIST_BACKEND=gloo and all ranks on the one GPU (RCCL refuses that): a functional check of the
    whole multi-rank path including the mailbox SyncBN and the time-out plumbing, not a measurement.  `--exec auto` (the driver's
    command) issues a multi-rank step eagerly; `--exec plan` replays the recorded step on every rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MGNET_DIST_BACKEND="gloo", MGNET_P2P_TIMEOUT_S="120")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--steps", "2" if gpus == 2 else "1", "--warmup", "1", "--batch", "2", "--height", "128",
           "--width", "256", "--no-cpu-baseline", "--timeout", "500", "--exec", how]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads(lines[-1])
    assert d["n_gpus"] == gpus and d["value"] > 0 and d["scaling"] == "weak", d
    dist_cfg = d["config"]["distributed"]
    if how == "auto":
        assert d["config"]["step_execution"].startswith("eager") and "--exec plan" in d["config"]["step_execution"], d["config"]["step_execution"]
    elif "p2p" in dist_cfg["syncbn_exchange"]["mode"]:
        assert d["config"]["step_execution"].startswith("plan"), d["config"]["step_execution"]
    assert d["config"]["host_issue_ms_per_step"] > 0
    assert dist_cfg["rccl_world_size"] == gpus and dist_cfg["grad_allreduce_calls_per_step"] >= 1, dist_cfg
    if "p2p" in dist_cfg["syncbn_exchange"]["mode"]:
        assert dist_cfg["syncbn_collectives_per_step"] == 0 and dist_cfg["syncbn_p2p_exchanges_per_step"] >= 100, dist_cfg
        assert dist_cfg["syncbn_exchange"]["wait_timed_out"] is False, dist_cfg
    assert all(v == v for v in d["config"]["losses"].values()), d["config"]["losses"]   # finite (NaN != NaN)


def _overlap_structure(rank, world):
    """how much of backward is still to be issued when the first bucket's all-reduce goes out: the reducer starts a bucket's collective from
    the hook of its last gradient (ready-order buckets), so most of backward's launches must come AFTER the first all_reduce call"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_network_cpu import small_model
    from mgnet_amd import _C
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer
    cfg, m = small_model(with_depth=True, seed=7)
    m = m.cuda()
    m.amp_dtype = torch.bfloat16
    tr = Trainer(cfg, m, bucket_bytes=1 << 20)    # (small model: small buckets, so that there are several)
    batch = synthetic_batch(1, 64, 96, "cuda", seed=100 + rank)
    for _ in range(2):
        tr.run_step(batch)
    calls = {"n": 0, "at_first_allreduce": None, "at_backward_start": None}
    real_stream = _C._stream

    def counting_stream():
        calls["n"] += 1
        return real_stream()
    _C._stream = counting_stream           # one call per library launch
    real_ar = dist.all_reduce

    def spy_all_reduce(*a, **k):
        if calls["at_first_allreduce"] is None and a and a[0].numel() > 64:
            calls["at_first_allreduce"] = calls["n"]
        return real_ar(*a, **k)
    dist.all_reduce = spy_all_reduce
    real_bw = tr._backward

    def spy_backward(loss_dict):
        calls["at_backward_start"] = calls["n"]
        return real_bw(loss_dict)
    tr._backward = spy_backward
    try:
        tr.run_step(batch)
    finally:
        _C._stream, dist.all_reduce = real_stream, real_ar
    torch.cuda.synchronize()
    bw_total = calls["n"] - calls["at_backward_start"]
    after = calls["n"] - calls["at_first_allreduce"]
    return len(tr.reducer.buckets), bw_total, after


def test_first_bucket_allreduce_is_issued_early_in_backward():
    out = _spawn(_overlap_structure)
    for r, (nb, bw_total, after) in out.items():
        assert nb >= 3, out
        assert after >= 0.6 * bw_total, out   # >= 60 % of backward's launches are issued after the first bucket's all_reduce call
