"""GPU: fused clip + Adam (mgnet_amd/csrc/optim.hip) against torch.optim.Adam + clip_grad_norm_ (the reference's
optimizer, tools/train_net.py:129-148) on identical parameters / gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("max_norm", [0.01, 0.0, 1e6])
def test_fused_adam_matches_torch(max_norm):
    from mgnet_amd import _C
    from mgnet_amd.engine import GradReducer
    from mgnet_amd.solver.fused_adam import FusedAdam

    torch.manual_seed(0)
    shapes = [(64, 3, 7, 7), (64,), (128, 64, 3, 3), (5,), (1000, 37), (1,), (256, 256, 1, 1)]
    ref_p = [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in shapes]
    my_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    lrs = [1e-3 if i % 2 else 1e-4 for i in range(len(shapes))]
    wds = [0.0 if i % 3 else 0.01 for i in range(len(shapes))]
    ref_opt = torch.optim.Adam([dict(params=[p], lr=lr, weight_decay=wd) for p, lr, wd in zip(ref_p, lrs, wds)], 1e-4)
    red = GradReducer(my_p, bucket_bytes=300_000, align=_C.optim_chunk(), flatten_params=True, average=False)
    assert len(red.buckets) >= 2
    my_opt = FusedAdam([dict(params=[p], lr=lr, weight_decay=wd) for p, lr, wd in zip(my_p, lrs, wds)], 1e-4, red,
                       max_grad_norm=max_norm)
    for step in range(6):
        my_opt.zero_grad()
        for p, q in zip(ref_p, my_p):
            g = torch.randn_like(p) * (10.0 ** (step % 3 - 1))
            p.grad = g.clone()
            q.grad = g.clone()
        red.finish()   # packs the handed-over gradients into the flat buckets (the trainer calls it before step())
        if max_norm > 0:
            total = torch.nn.utils.clip_grad_norm_(ref_p, max_norm)
        ref_opt.step()
        my_opt.step()
        if max_norm > 0:
            assert float(my_opt.grad_norm()) == pytest.approx(float(total), rel=1e-5)
        for g in ref_opt.param_groups + my_opt.param_groups:  # an LR schedule
            g["lr"] *= 0.9
    for p, q in zip(ref_p, my_p):
        assert torch.allclose(p, q, rtol=2e-5, atol=1e-7), float((p - q).abs().max())


def test_per_step_tables_survive_a_host_that_runs_ahead():
    """The LR / weight-decay tables go to the device through pinned staging buffers while the host is not synchronised with
    the GPU: with the queue kept busy, six steps with a halving LR are issued before the first one executes.  (A staging
    buffer rewritten before its copy ran would apply a later step's LR to an earlier step.)"""
    from mgnet_amd import _C
    from mgnet_amd.engine import GradReducer
    from mgnet_amd.solver.fused_adam import FusedAdam

    torch.manual_seed(1)
    ref_p = [torch.nn.Parameter(torch.randn(300, 300, device="cuda")), torch.nn.Parameter(torch.randn(77, device="cuda"))]
    my_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    grads = [[torch.randn_like(p) for p in ref_p] for _ in range(6)]
    ref_opt = torch.optim.Adam([dict(params=[p], lr=1e-2) for p in ref_p], 1e-2)
    red = GradReducer(my_p, bucket_bytes=1 << 20, align=_C.optim_chunk(), flatten_params=True, average=False)
    my_opt = FusedAdam([dict(params=[p], lr=1e-2) for p in my_p], 1e-2, red)
    for step in range(6):
        for p, g in zip(ref_p, grads[step]):
            p.grad = g.clone()
        ref_opt.step()
        for g in ref_opt.param_groups:
            g["lr"] *= 0.5
    torch.cuda.synchronize()
    busy = torch.randn(8192, 8192, device="cuda")
    for _ in range(40):   # ~0.3 s of queued work
        busy = (busy @ busy) * 1e-4
    for step in range(6):
        my_opt.zero_grad()
        for q, g in zip(my_p, grads[step]):
            q.grad = g.clone()
        red.finish()
        my_opt.step()
        for g in my_opt.param_groups:
            g["lr"] *= 0.5
    torch.cuda.synchronize()
    for p, q in zip(ref_p, my_p):
        assert torch.allclose(p, q, rtol=2e-5, atol=1e-6), float((p - q).abs().max())


def test_grad_scale_of_a_two_rank_world():
    """With average=False the reducer hands the SUM over ranks to the optimizer, which folds 1/world into the clipping pass:
    a summed gradient of 2 g with world = 2 must give the update of g with world = 1 (same clipping decision, same Adam moments)."""
    from mgnet_amd import _C
    from mgnet_amd.engine import GradReducer
    from mgnet_amd.solver.fused_adam import FusedAdam

    def run(world, scale):
        torch.manual_seed(3)
        ps = [torch.nn.Parameter(torch.randn(128, 64, 3, 3, device="cuda")), torch.nn.Parameter(torch.randn(64, device="cuda"))]
        red = GradReducer(ps, bucket_bytes=1 << 20, align=_C.optim_chunk(), flatten_params=True, average=False)
        red.world = world
        opt = FusedAdam([dict(params=[p]) for p in ps], 1e-3, red, max_grad_norm=0.01)
        gs = [[torch.randn_like(p) for p in ps] for _ in range(3)]
        for step in range(3):
            opt.zero_grad()
            for p, g in zip(ps, gs[step]):
                p.grad = g * scale
            red.world = 1          # (no process group here: pack without launching collectives)
            red.finish()
            red.world = world
            opt.step()
        return [p.detach().clone() for p in ps], float(opt.grad_norm())

    a, na = run(1, 1.0)
    b, nb = run(2, 2.0)
    assert na == pytest.approx(nb, rel=1e-6)
    for x, y in zip(a, b):
        assert torch.allclose(x, y, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("kind,nesterov", [("ADAMW", False), ("SGD", False), ("SGD", True)])
def test_fused_adamw_and_sgd_match_torch(kind, nesterov):
    """the other two optimizers tools/train_net.py:129-154 can build (SOLVER.OPTIMIZER "ADAMW" / "SGD"): fused clip + step on the flat
    buckets against torch.optim.AdamW / torch.optim.SGD(momentum, nesterov) + clip_grad_norm_, per-group lr and weight decay, an LR schedule"""
    from mgnet_amd import _C
    from mgnet_amd.engine import GradReducer
    from mgnet_amd.solver.fused_adam import FusedAdam

    torch.manual_seed(4)
    shapes = [(64, 3, 7, 7), (64,), (128, 64, 3, 3), (5,), (1000, 37), (1,)]
    ref_p = [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in shapes]
    my_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    lrs = [1e-2 if i % 2 else 1e-3 for i in range(len(shapes))]
    wds = [0.0 if i % 3 else 0.05 for i in range(len(shapes))]
    groups = lambda ps: [dict(params=[p], lr=lr, weight_decay=wd) for p, lr, wd in zip(ps, lrs, wds)]
    if kind == "ADAMW":
        ref_opt = torch.optim.AdamW(groups(ref_p), 1e-3)
    else:
        ref_opt = torch.optim.SGD(groups(ref_p), 1e-3, momentum=0.9, nesterov=nesterov)
    red = GradReducer(my_p, bucket_bytes=300_000, align=_C.optim_chunk(), flatten_params=True, average=False)
    my_opt = FusedAdam(groups(my_p), 1e-3, red, max_grad_norm=0.5, kind=kind, momentum=0.9, nesterov=nesterov)
    for step in range(6):
        my_opt.zero_grad()
        for p, q in zip(ref_p, my_p):
            g = torch.randn_like(p) * (10.0 ** (step % 3 - 1))
            p.grad, q.grad = g.clone(), g.clone()
        red.finish()
        torch.nn.utils.clip_grad_norm_(ref_p, 0.5)
        ref_opt.step()
        my_opt.step()
        for g in ref_opt.param_groups + my_opt.param_groups:
            g["lr"] *= 0.9
    for p, q in zip(ref_p, my_p):
        assert torch.allclose(p, q, rtol=3e-5, atol=2e-7), (kind, float((p - q).abs().max()))
    sd = my_opt.state_dict()
    assert ("momentum_buffer" in sd["state"][0]) == (kind == "SGD")
