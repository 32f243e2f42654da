"""GPU: the kernels of multi-scale + flip inference (csrc/mscflip.hip) against the torch formulation of
mg_net.py:427-520 -- F.interpolate(bilinear, align_corners=True), softmax, offset scaling, inv2depth, torch.flip, running sums."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_pass(lr, mode, flip, stride, scale):
    up = F.interpolate(lr.float(), scale_factor=stride / scale, mode="bilinear", align_corners=True)
    if mode == "softmax":
        v = torch.softmax(up, 1)
    elif mode == "offset":
        v = up * stride / scale
    elif mode == "inv2depth":
        v = 1.0 / up.clamp(min=1e-6)
    else:
        v = up
    if flip:
        v = torch.flip(v, dims=(3,))
        if mode == "offset":
            v[:, 1] *= -1
    return v


@pytest.mark.parametrize("mode,C,dtype", [("softmax", 20, torch.bfloat16), ("softmax", 19, torch.float16), ("plain", 1, torch.float32),
                                          ("offset", 2, torch.bfloat16), ("inv2depth", 1, torch.float32)])
def test_accumulate_matches_torch_over_all_passes(mode, C, dtype):
    """the seven scales x two flips of the reference's default schedule, on head outputs of the sizes those passes produce (ragged:
    H=96, W=160 -> stride-8 maps of 6x10 ... 24x40), strided like the predictors' channel-padded outputs"""
    from mgnet_amd import _C
    torch.manual_seed(0)
    N, H, W, stride = 2, 96, 160, 8
    scales = [0.5, 0.75, 1.0, 1.25, 1.5, 1.75, 2.0]
    acc = torch.empty(N, C, H, W, device="cuda")
    ref = None
    n = 2 * len(scales)
    k = 0
    for scale in scales:
        h, w = int(math.floor(H * scale)) // stride, int(math.floor(W * scale)) // stride
        for f in range(2):
            pad = torch.randn(N, 32, h, w, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
            lr = pad[:, :C] if dtype != torch.float32 else (torch.rand(N, C, h, w, device="cuda") * 1.9 + 0.05 if mode == "inv2depth"
                                                            else torch.rand(N, C, h, w, device="cuda") * 2 - 1.0)
            assert tuple(F.interpolate(lr.float(), scale_factor=stride / scale, mode="bilinear", align_corners=True).shape[2:]) == (H, W)
            _C.msc_accumulate(acc, lr, mode, f, k == 0, stride=float(stride), scale=float(scale), divide=float(n) if k == n - 1 else 0.0)
            v = _ref_pass(lr, mode, f, stride, scale)
            ref = v if ref is None else ref + v
            k += 1
    ref = ref / n
    # fp32 sums of 14 terms evaluated in two orders of association: a few 1e-7 of the largest term (the averages of mixed-sign maps pass
    # through zero, so the bound is relative to the map's scale, not per element; 1 / depth is relative per element)
    if mode == "inv2depth":
        err = float(((acc - ref).abs() / ref.abs()).max())
        assert err < 2e-5, (mode, err)
    else:
        err = float((acc - ref).abs().max() / ref.abs().max())
        assert err < 3e-6, (mode, err)
    if mode == "softmax":
        assert torch.allclose(acc.sum(1), torch.ones(N, H, W, device="cuda"), atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_input_rescale_flip_matches_torch(dtype):
    from mgnet_amd import _C
    torch.manual_seed(1)
    N, H, W = 2, 70, 122
    norm = torch.randn(N, 3, H, W, device="cuda")
    for scale in (0.5, 0.75, 1.0, 1.25, 2.0):
        x = F.interpolate(norm, scale_factor=scale, mode="bilinear", align_corners=True)
        h, w = x.shape[2:]
        assert (h, w) == (int(math.floor(H * scale)), int(math.floor(W * scale)))
        for f in (0, 1):
            want = torch.flip(x, dims=(3,)) if f else x
            got = _C.msc_input(norm, h, w, f, dtype)
            assert got.shape == (N, 8, h, w) and got.is_contiguous(memory_format=torch.channels_last)
            assert float(got[:, 3:].abs().max()) == 0.0
            # same fp32 interpolation up to contraction order, then ONE rounding to 16 bits: at most one 16-bit ulp apart, rarely
            d = (got[:, :3].float() - want.to(dtype).float()).abs()
            # (spacing of the 16-bit grid at |x|: between 2^-8 |x| and 2^-7 |x| for bf16, 2^-11 .. 2^-10 for fp16)
            ulp = (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10) * want.abs().clamp(min=2.0 ** -14)
            # (+ the fp32 round-off of the interpolation itself, which is what remains where the blend of mixed-sign pixels passes through zero)
            assert bool((d <= ulp * 1.01 + 4e-6).all()) and float((d > 0).float().mean()) < 2e-2, (scale, f, float(d.max()))


def test_input_rescale_flip_fp32_matches_torch():
    """the fp32 trunk's network input (SOLVER.AMP.ENABLED False): [N,3,h,w] fp32 channels_last, two fp32 evaluation orders apart"""
    from mgnet_amd import _C
    torch.manual_seed(2)
    N, H, W = 2, 70, 122
    norm = torch.randn(N, 3, H, W, device="cuda")
    for scale in (0.5, 1.0, 1.75):
        x = F.interpolate(norm, scale_factor=scale, mode="bilinear", align_corners=True)
        h, w = x.shape[2:]
        for f in (0, 1):
            want = torch.flip(x, dims=(3,)) if f else x
            got = _C.msc_input(norm, h, w, f, torch.float32)
            assert got.shape == (N, 3, h, w) and got.dtype == torch.float32 and got.is_contiguous(memory_format=torch.channels_last)
            assert float((got - want).abs().max()) < 4e-6, (scale, f)


def test_hip_path_matches_torch_formulation_end_to_end():
    """forward_multi_scale_flip on the bf16 HIP trunk: device path (mscflip.hip) vs the torch formulation driving the SAME network"""
    from test_model_golden import GM, _model
    m = _model("cuda", True).eval()
    img = GM.eval_image().cuda()
    with torch.no_grad():
        x = ((img[None].float() / 255.0) - m.pixel_mean) / m.pixel_std
        got = m.forward_multi_scale_flip(x, scales=[0.5, 1.0, 1.5], flip=True)
        want = m.forward_multi_scale_flip(x, scales=[0.5, 1.0, 1.5], flip=True, _torch_formulation=True)   # the same network, torch algebra
    # the two paths feed the network inputs that differ in the last bf16 bit at a few pixels (two fp32 evaluation orders of the rescale):
    # the outputs agree to bf16 noise almost everywhere, a soft-max probability next to a near-tie of two logits moves further
    for k in ("sem_seg", "center", "offset"):
        a, b = got[k].float(), want[k].float()
        assert a.shape == b.shape
        d, scale = (a - b).abs(), float(b.abs().max())
        assert float((d > 2e-2 * scale + 1e-4).float().mean()) < 2e-3 and float(d.max()) <= 0.15 * scale, (k, float(d.max()), scale)
    a, b = 1.0 / got["depth"], 1.0 / want["depth"]
    assert float((a - b).abs().max()) < 0.05
