"""GPU: the fp16 activation path (the reference's AMP format, BASELINE C5: `x_f16.hip` twins of every kernel that touches
activations + GradScaler-style dynamic loss scaling on the device) against the same oracles as the bf16 path."""
import numpy as np
import pytest
import torch

from test_network_cpu import small_model
from test_network_gpu import _randomise

pytestmark = pytest.mark.gpu


def test_conv_fp16_matches_fp64_reference():
    """implicit-GEMM forward / data gradient / weight gradient on v_mfma_f32_32x32x16_f16 vs fp64 F.conv2d"""
    from mgnet_amd import _C
    from mgnet_amd.modeling import ops
    torch.manual_seed(0)
    for (cin, cout, k, s, hw) in [(64, 64, 3, 1, (40, 72)), (128, 256, 3, 2, (32, 64)), (256, 256, 1, 1, (24, 40)), (256, 19, 1, 1, (16, 24))]:
        x = torch.randn(2, cin, *hw, device="cuda").half().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = torch.nn.Parameter(torch.randn(cout, cin, k, k, device="cuda") * (2.0 / (cin * k * k)) ** 0.5)
        y = ops.conv2d(x, w, None, stride=s, padding=k // 2)
        assert y.dtype == torch.float16
        g = torch.randn_like(y)
        y.backward(g)
        xd, wd = x.detach().double().requires_grad_(True), w.detach().half().double().requires_grad_(True)
        yr = torch.nn.functional.conv2d(xd, wd, None, stride=s, padding=k // 2)
        yr.backward(g.double())
        rel = lambda a, b: float((a.double() - b).abs().max() / (b.abs().max() + 1e-12))
        assert rel(y, yr) < 4e-3, (cin, cout, k, s, rel(y, yr))          # fp16: 11-bit mantissa, fp32 accumulation
        assert rel(x.grad, xd.grad) < 4e-3, (cin, cout, k, s, rel(x.grad, xd.grad))
        assert rel(w.grad, wd.grad) < 4e-3, (cin, cout, k, s, rel(w.grad, wd.grad))


def test_iabn_and_eltwise_fp16():
    from mgnet_amd.modeling import ops
    torch.manual_seed(1)
    x = (torch.randn(4, 64, 24, 40, device="cuda") * 3 + 1).half().contiguous(memory_format=torch.channels_last)
    w, b = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1
    rm, rv = torch.zeros(64, device="cuda"), torch.ones(64, device="cuda")
    xin = x.clone().requires_grad_(True)
    wp, bp = torch.nn.Parameter(w.clone()), torch.nn.Parameter(b.clone())
    y = ops.iabn(xin * 1.0, wp, bp, rm, rv, True, 0.01, 1e-5, "leaky_relu", 0.01)
    g = torch.randn_like(y)
    y.backward(g)
    xd = x.double().requires_grad_(True)
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = torch.nn.functional.leaky_relu(torch.nn.functional.batch_norm(xd, None, None, wd.abs() + 1e-5, bd, True, 0.0, 1e-5), 0.01)
    yr.backward(g.double())
    assert float((y.double() - yr).abs().max()) < 5e-3 * float(yr.abs().max())
    assert float((xin.grad.double() - xd.grad).abs().max()) < 2e-2 * float(xd.grad.abs().max())
    assert float((wp.grad.double() - wd.grad * torch.sign(wd)).abs().max()) < 2e-2 * float(wd.grad.abs().max())
    a = torch.randn(2, 64, 8, 16, device="cuda").half().contiguous(memory_format=torch.channels_last)
    c = torch.randn(2, 64, 8, 16, device="cuda").half().contiguous(memory_format=torch.channels_last)
    assert torch.equal(ops.add_relu(a, c), torch.relu(a.float() + c.float()).half())
    assert ops.max_pool_3x3_s2(a).dtype == torch.float16
    assert torch.equal(ops.max_pool_3x3_s2(a), torch.nn.functional.max_pool2d(a, 3, 2, 1))


def test_full_step_fp16_matches_oracle_and_trains():
    """losses of the fp16 network vs the fp32 oracle (rel 2e-2), then optimizer steps with the dynamic loss scale: the scale
    backs off on overflow, steps are skipped exactly then, and the loss goes down"""
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer
    from oracle import network_oracle as NO

    cfg, m = small_model(with_depth=True, seed=3)
    _randomise(m)
    m.train()
    batch = synthetic_batch(2, 128, 192, "cpu", seed=5)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ref = NO.mgnet_losses(sd, batch, pixel_mean=cfg.MODEL.PIXEL_MEAN, pixel_std=cfg.MODEL.PIXEL_STD, ohem_n_min=1500)
    m = m.cuda()
    m.amp_dtype = torch.float16
    dev_batch = [{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in x.items()} for x in batch]
    got = m(dev_batch)
    for k in ref:
        assert float(got[k]) == pytest.approx(float(ref[k]), rel=2e-2, abs=2e-4), k
    tr = Trainer(cfg, m)
    assert tr.optimizer.scaler is not None and float(tr.optimizer.scaler[0]) == 65536.0
    tot, scales, steps = [], [], []
    for _ in range(12):
        out = tr.run_step(dev_batch)
        tot.append(float(sum(v.detach() for v in out.values())))
        scales.append(float(tr.optimizer.scaler[0]))
        steps.append(float(tr.optimizer.scaler[2]))
    assert all(np.isfinite(tot)) and tot[-1] < tot[0], tot
    assert steps[-1] >= 6, (steps, scales)                      # most steps are taken
    for i in range(1, len(steps)):                              # a skipped step <=> the scale was halved
        assert (steps[i] == steps[i - 1]) == (scales[i] < scales[i - 1]), (steps, scales)
    for p in m.parameters():
        assert bool(torch.isfinite(p).all())


def test_fp16_full_size_steps_keep_the_loss_scale():
    """BASELINE C5 at its own resolution (1024 x 2048, 4 frames): fp16's 65504 range is a real limit for the full-resolution stems, so ten
    optimizer steps must stay finite, the dynamic loss scale must not collapse (>= 2^8: at most eight back-offs from 2^16) and most steps
    must be taken."""
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer

    cfg, m = small_model(with_depth=True, seed=11)
    m = m.cuda().train()
    m.amp_dtype = torch.float16
    tr = Trainer(cfg, m)
    batch = synthetic_batch(4, 1024, 2048, "cuda", seed=21)
    tot, scales = [], []
    for _ in range(10):
        out = tr.run_step(batch)
        tot.append(float(sum(v.detach() for v in out.values())))
        scales.append(float(tr.optimizer.scaler[0]))
    steps_taken = float(tr.optimizer.scaler[2])
    assert all(np.isfinite(tot)), tot
    assert min(scales) >= 2.0 ** 8, scales
    assert steps_taken >= 6, (steps_taken, scales)
    for n, p in m.named_parameters():
        assert bool(torch.isfinite(p).all()), n
