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
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, fn(rank, world)))
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
        p.join(120)
        assert p.exitcode == 0
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
