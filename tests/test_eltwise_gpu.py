"""GPU: element-wise / broadcast / pooling glue kernels (csrc/eltwise.hip) against the torch formulation of the reference
lines they replace, on the same bf16 values."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _x(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(torch.bfloat16)


def _pair(x0):
    return x0.float().requires_grad_(True), x0.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)


def _close(a, b, tol=2e-2):
    return float((a.detach().float().cpu() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-9)) < tol


@pytest.mark.parametrize("shape", [(2, 64, 9, 13), (1, 512, 4, 6)])
def test_add_relu(shape):
    from mgnet_amd.modeling import ops
    (ar, a), (br, b) = _pair(_x(shape, 1)), _pair(_x(shape, 2))
    g = _x(shape, 3)
    yr = F.relu(ar + br); (yr * g.float()).sum().backward()
    y = ops.add_relu(a, b); (y.float() * g.cuda().float()).sum().backward()
    assert _close(y, yr) and _close(a.grad, ar.grad) and _close(b.grad, br.grad)


@pytest.mark.parametrize("shape", [(2, 128, 32, 64), (3, 256, 7, 5), (2, 512, 1, 1)])
def test_global_avg_pool(shape):
    from mgnet_amd.modeling import ops
    xr, x = _pair(_x(shape, 4))
    g = _x(shape[:2] + (1, 1), 5)
    yr = xr.mean((2, 3), keepdim=True); (yr * g.float()).sum().backward()
    y = ops.global_avg_pool(x); (y.float() * g.cuda().float()).sum().backward()
    assert y.shape == yr.shape and _close(y, yr) and _close(x.grad, xr.grad)


@pytest.mark.parametrize("cfg", [((2, 128, 4, 6), (8, 12)), ((1, 64, 1, 1), (5, 7)), ((2, 128, 3, 5), (7, 11))])
def test_nearest_upsample(cfg):
    from mgnet_amd.modeling import ops
    shape, size = cfg
    xr, x = _pair(_x(shape, 6))
    yr = F.interpolate(xr, size=size, mode="nearest")
    g = _x(tuple(yr.shape), 7)
    (yr * g.float()).sum().backward()
    y = ops.upsample_nearest(x, size); (y.float() * g.cuda().float()).sum().backward()
    assert torch.equal(y.float().cpu(), yr.detach()) and _close(x.grad, xr.grad)


@pytest.mark.parametrize("residual", [False, True])
def test_scale_channels(residual):
    from mgnet_amd.modeling import ops
    shape = (2, 128, 6, 10)
    xr, x = _pair(_x(shape, 8))
    s0 = torch.sigmoid(_x((2, 128, 1, 1), 9).float())
    sr, s = s0.clone().requires_grad_(True), s0.cuda().requires_grad_(True)
    g = _x(shape, 10)
    yr = xr + xr * sr if residual else xr * sr
    (yr * g.float()).sum().backward()
    y = ops.scale_channels(x, s, residual=residual); (y.float() * g.cuda().float()).sum().backward()
    assert _close(y, yr) and _close(x.grad, xr.grad) and _close(s.grad, sr.grad)


def test_concat_channels():
    from mgnet_amd.modeling import ops
    (ar, a), (br, b) = _pair(_x((2, 128, 5, 7), 11)), _pair(_x((2, 64, 5, 7), 12))
    yr = torch.cat([ar, br], 1)
    g = _x(tuple(yr.shape), 13)
    (yr * g.float()).sum().backward()
    y = ops.concat_channels(a, b); (y.float() * g.cuda().float()).sum().backward()
    assert torch.equal(y.float().cpu(), yr.detach()) and torch.equal(a.grad.float().cpu(), ar.grad) and torch.equal(b.grad.float().cpu(), br.grad)


def test_u8_frames_to_f32_is_exactly_float_div_255():
    """mgn_u8_frames_to_f32 (stack + `.float() / 255` of the un-jittered frames, mg_net.py:320-335): bit-identical values"""
    import torch
    from mgnet_amd import _C
    torch.manual_seed(0)
    frames = [torch.randint(0, 256, (3, 64, 96), dtype=torch.uint8, device="cuda") for _ in range(5)]
    frames[1][0, 0, :16] = torch.arange(240, 256, dtype=torch.uint8, device="cuda")
    out = _C.u8_frames_to_f32(frames, 255.0)
    assert out is not None and tuple(out.shape) == (5, 3, 64, 96)
    # IEEE division like the reference's CPU path / numpy (torch's GPU kernel multiplies by the reciprocal of a scalar divisor,
    # which differs by one ulp for some byte values)
    import numpy as np
    want = torch.stack(frames).cpu().numpy().astype(np.float32) / np.float32(255.0)
    assert np.array_equal(out.cpu().numpy(), want)
    allv = torch.arange(256, dtype=torch.uint8, device="cuda")   # every byte value
    assert np.array_equal(_C.u8_frames_to_f32([allv], 255.0)[0].cpu().numpy(), np.arange(256, dtype=np.float32) / np.float32(255.0))
    assert torch.allclose(out, torch.stack(frames).float() / 255.0, rtol=2e-7, atol=0)
    # unsupported layouts fall back (None): not a multiple of 16 bytes, non-contiguous
    assert _C.u8_frames_to_f32([torch.zeros(3, 5, 7, dtype=torch.uint8, device="cuda")], 255.0) is None
    assert _C.u8_frames_to_f32([frames[0].transpose(1, 2)], 255.0) is None


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fanout3_sums_the_three_gradients_in_one_pass(dtype):
    """ops.fanout3 (backbone features -> three heads): three aliases in the forward; the backward is a + b + c in fp32 with one rounding
    (csrc/eltwise.hip sum3), also with one branch unused (two gradients) and for a layout the kernel does not take (fallback)."""
    from mgnet_amd import _C
    from mgnet_amd.modeling import ops

    torch.manual_seed(0)
    x = torch.randn(2, 64, 12, 20, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    a, b, c = ops.fanout3(x)
    assert a.data_ptr() == x.data_ptr() and torch.equal(a, x) and torch.equal(c, x)
    g = [torch.randn_like(x) for _ in range(3)]
    (a.float() * g[0].float()).sum().backward(retain_graph=True)
    assert torch.equal(x.grad, g[0])                                                     # one gradient: passed through
    x.grad = None
    ((a.float() * g[0].float()).sum() + (b.float() * g[1].float()).sum() + (c.float() * g[2].float()).sum()).backward()
    ref = ((g[0].float() + g[1].float()) + g[2].float()).to(dtype)
    assert torch.equal(x.grad, ref)
    assert torch.equal(_C.sum3(g[0], g[1]), (g[0].float() + g[1].float()).to(dtype))
    # a shape the 16-byte kernel does not take (numel % 8 != 0): torch fallback, same value up to the intermediate rounding
    y = torch.randn(1, 3, 5, 7, device="cuda").to(dtype).requires_grad_(True)
    u, v, w = ops.fanout3(y)
    (u.float().sum() + 2 * v.float().sum() + 3 * w.float().sum()).backward()
    assert torch.allclose(y.grad.float(), torch.full_like(y, 6.0).float())
