"""CPU, world_size 2 over gloo: the data-parallel pieces of the step -- cross-rank activated-batch-norm
statistics and the bucketed gradient reducer."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, fn(rank, world)))
    except Exception as e:  # report instead of leaving the parent blocked on the queue
        import traceback
        q.put((rank, "worker failed: " + "".join(traceback.format_exception(type(e), e, e.__traceback__))))
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
        p.join(60)
    for r, v in out.items():
        assert v is True, (r, v)
    for p in procs:
        assert p.exitcode == 0
    return out


def _syncbn(rank, world):
    import torch.nn.functional as F
    from mgnet_amd.modeling import ops

    torch.manual_seed(0)
    full = torch.randn(4, 6, 5, 7) * 2 + 1
    w, b = torch.rand(6) + 0.5, torch.randn(6) * 0.1
    g_full = torch.randn(4, 6, 5, 7)
    x = full[rank * 2:(rank + 1) * 2].clone().requires_grad_(True)
    wl, bl = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rm, rv = torch.zeros(6), torch.ones(6)
    y = ops.iabn(x, wl, bl, rm, rv, True, 0.01, 1e-5, "leaky_relu", 0.01, group=dist.group.WORLD)
    (y * g_full[rank * 2:(rank + 1) * 2]).sum().backward()
    # single-process truth on the whole batch
    xf = full.clone().requires_grad_(True)
    wf, bf = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yf = F.leaky_relu(F.batch_norm(xf, None, None, wf.abs() + 1e-5, bf, True, 0.0, 1e-5), 0.01)
    (yf * g_full).sum().backward()
    sl = slice(rank * 2, (rank + 1) * 2)
    ok = torch.allclose(y, yf[sl], atol=1e-5) and torch.allclose(x.grad, xf.grad[sl], atol=1e-5)
    # parameter grads are per-rank partial sums: their all-reduce equals the full-batch gradient
    gw = wl.grad.clone(); dist.all_reduce(gw)
    ok = ok and torch.allclose(gw, wf.grad, atol=1e-4)
    n = full.numel() / 6
    ok = ok and torch.allclose(rv, 0.99 * torch.ones(6) + 0.01 * full.var((0, 2, 3), unbiased=False) * n / (n - 1), atol=1e-5)
    return bool(ok)


def test_syncbn_statistics_world2():
    assert all(_spawn(_syncbn).values())


def _reducer(rank, world):
    from mgnet_amd.engine import GradReducer

    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 1000))
    import copy
    ref = copy.deepcopy(net)             # same weights, no reducer: the per-rank local gradients
    params = list(net.parameters())
    red = GradReducer(params, bucket_bytes=4096)  # several buckets
    assert len(red.buckets) >= 3
    torch.manual_seed(10 + rank)
    x = torch.randn(5, 8)
    red.zero_grad()
    net[:3](x).sum().backward()      # the last layer gets no gradient this step (unused-parameter path)
    ref[:3](x).sum().backward()
    local = [torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in ref.parameters()]
    red.finish()
    ok0 = all(p.grad.data_ptr() == b["flat_g"].data_ptr() + 4 * o          # .grad are views into the flat buckets again
              for b in red.buckets for p, o in zip(b["params"], b["offsets"]))
    gathered = [[torch.zeros_like(g) for _ in range(world)] for g in local]
    for lst, g in zip(gathered, local):
        dist.all_gather(lst, g)
    ok = ok0 and all(torch.allclose(p.grad, sum(lst) / world, atol=1e-6) for p, lst in zip(params, gathered))
    ok = ok and float(params[-1].grad.abs().max()) == 0.0
    # second step reuses the same flat buffers
    red.zero_grad()
    net(x).sum().backward()
    red.finish()
    ok = ok and float(params[-1].grad.abs().max()) > 0.0
    return bool(ok)


def test_grad_reducer_world2():
    assert all(_spawn(_reducer).values())
