"""GPU: the COMPOSED bf16 / fp16 training step against a reference that rounds where the HIP path rounds.

tests/test_grad_parity_gpu.py compares every parameter gradient of the 16-bit step with the fp32 oracle and is bounded from below by
what 16-bit storage costs this network (about 60 norm layers amplify every rounding; a plain-torch bf16 run is 0.4 away as well), so
it cannot be sharp.  tests/test_step_nodes_gpu.py is sharp but node-local.  This file closes the gap between them: the same model,
weights and batch are stepped twice --

  (a) on the product path: HIP convolutions (windowed / implicit-GEMM / stem kernels, shortcut gradient inside the data-gradient
      kernel, channel-padded predictors, deferred split-K weight gradients), batch statistics from the conv epilogues, in-place
      InPlaceABN with its backward re-derived from the stored output, the fused stem and block-tail forms;
  (b) with every `_ConvFn`, `_IABNFn`, `_AbnAddReluFn`, `_AbnPoolFn` node evaluated by fp64 torch expressions of the reference's
      operators (detectron2 Conv2d = conv -> norm -> activation, res_net.py:35-79; inplace_abn's batch norm with gamma = |weight| + eps
      and its from-the-output backward) that ROUND TO 16 BITS AT THE SAME POINTS: conv output, norm output, (norm output) + shortcut,
      every data gradient.  All other nodes are the same in both runs.

The review of round 2 expected the residual between (a) and (b) to be fp32-vs-fp64 accumulation only, hence cosine >= 0.999 per
tensor.  MEASURED, it is not: (a) vs (b) has a median per-tensor cosine of 0.94-0.95 in bf16 (relative error 0.31-0.35), 0.991 in fp16
(0.13) -- and so has run (c), the control: THE SAME substituted nodes evaluated in fp32 instead of fp64 arithmetic (0.91-0.95 / 0.31-0.42;
fp16 0.990 / 0.14).  Rounding is discontinuous: a difference of 1e-7 between two accumulations flips a share ~1e-7 / 2^-8 of the
roundings of a tensor, each flip is a whole ulp, the next layer's roundings turn that sparse field into a dense one of a fraction of
an ulp, and after a few layers two evaluations that round at the same points carry independent rounding noise -- which the ~60 norm
backward passes then amplify exactly like the noise of a plain bf16 evaluation.  That the error scales with the format's ulp (fp16: 8x
finer, error ~3x smaller = sqrt(8)) and not with the accumulation precision is the signature.  No rounding-matched oracle of this
network reaches 0.999; the sharp statement about the composed step is therefore the node-local one (test_step_nodes_gpu.py: every
node from its own recorded inputs, <= 0.7 ulp of a tensor norm), and what this file asserts is that the product path is
INDISTINGUISHABLE from the fp32 twin of the rounding-matched reference: losses within 3x the control's distance, per-tensor gradient
errors no worse than the control's in aggregate and none far off (a wrong fused term in any conv / norm node -- they carry > 95 % of
the step's arithmetic -- is an O(1) difference in every tensor upstream of it, in every precision)."""
import os

import pytest
import torch
import torch.nn.functional as F

from test_grad_parity_gpu import _rows, _significant
from test_network_cpu import small_model
from test_network_gpu import _randomise

pytestmark = pytest.mark.gpu


ACC = torch.float64     # arithmetic of the substituted nodes (the control run uses fp32: same rounding points, other accumulation)


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _ste_round(y, dtype):
    """value rounded to `dtype`, gradient of the unrounded expression (the stored tensor is 16-bit, its adjoint passes through)"""
    return y + (y.detach().to(dtype).to(ACC) - y.detach())


class _RefConv:
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, relu, with_skip=False, cout_pad=0, stats=None, keep_pad=False):
        cout, cin = weight.shape[:2]
        dt = x.dtype
        y = F.conv2d(x[:, :cin].to(ACC), weight.detach().to(dt).to(ACC), None if bias is None else bias.detach().to(ACC),
                     stride=stride, padding=pad)
        if relu:
            y = torch.relu(y)
        out = y.to(dt)
        if cout_pad:
            out = F.pad(out, (0, 0, 0, 0, 0, cout_pad - cout))
        out = _cl(out)
        ctx.save_for_backward(x, weight, bias, out if relu else None)
        ctx.cfg = (stride, pad, relu, cout_pad, keep_pad)
        if cout_pad and keep_pad:
            return out
        if cout_pad:
            return out[:, :cout]
        if with_skip == 2:     # (down-sampling block: the shortcut conv's differentiable input, see ops._ShortcutS2Fn)
            return out, x[:, :, ::2, ::2]
        if with_skip:
            return out, x
        return out

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, weight, bias, out = ctx.saved_tensors
        stride, pad, relu, cout_pad, keep_pad = ctx.cfg
        cout, cin = weight.shape[:2]
        dt = x.dtype
        g = dy.to(dt)[:, :cout].to(ACC)
        if relu:
            g = g * (out[:, :cout] > 0)
        need = ctx.needs_input_grad
        with torch.enable_grad():
            xd = x[:, :cin].to(ACC).requires_grad_(bool(need[0]))
            wd = weight.detach().to(dt).to(ACC).requires_grad_(True)
            bd = None if bias is None else bias.detach().to(ACC).requires_grad_(True)
            y = F.conv2d(xd, wd, bd, stride=stride, padding=pad)
        wanted = [t for t in (xd if need[0] else None, wd, bd) if t is not None]
        got = dict(zip([id(t) for t in wanted], torch.autograd.grad(y, wanted, g)))
        dx = None
        if need[0]:
            dx = got[id(xd)]
            if dskip is not None and dskip.shape != dx.shape:   # with_skip = 2: the shortcut's gradient belongs to the even pixels
                dx = dx.clone()
                dx[:, :, ::2, ::2] += dskip.to(dt).to(ACC)
            elif dskip is not None:
                dx = dx + dskip.to(dt).to(ACC)        # the shortcut gradient joins before the one rounding of the kernel's epilogue
            dx = _cl(dx.to(dt))
        dw = got[id(wd)].to(weight.dtype) if need[1] else None
        db = got[id(bd)].to(bias.dtype) if bd is not None and need[2] else None
        return dx, dw, db, None, None, None, None, None, None, None


class _RefShortcutS2:
    """ops._ShortcutS2Fn: 1x1 / stride 2 / pad 0 conv of `xfull`, its data gradient handed to `xsub` at the low resolution"""

    @staticmethod
    def forward(ctx, xsub, xfull, weight, stats):
        dt = xfull.dtype
        y = F.conv2d(xfull.to(ACC), weight.detach().to(dt).to(ACC), None, stride=2, padding=0)
        ctx.save_for_backward(xfull, weight)
        return _cl(y.to(dt))

    @staticmethod
    def backward(ctx, dy):
        xfull, weight = ctx.saved_tensors
        dt = xfull.dtype
        with torch.enable_grad():
            xd = xfull.to(ACC).requires_grad_(True)
            wd = weight.detach().to(dt).to(ACC).requires_grad_(True)
            y = F.conv2d(xd, wd, None, stride=2, padding=0)
        dx, dw = torch.autograd.grad(y, (xd, wd), dy.to(dt).to(ACC))
        return _cl(dx[:, :, ::2, ::2].to(dt)), None, dw.to(weight.dtype), None


def _stats(xd):
    return xd.mean((0, 2, 3), keepdim=True), xd.var((0, 2, 3), unbiased=False, keepdim=True)


def _shape(v):
    return v.to(ACC).view(1, -1, 1, 1)


class _RefIABN:
    """inplace_abn: y = act(gamma * x_hat + beta) written over x; backward from y alone (z = act^-1(y), x_hat = (z - beta) / gamma)"""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, activation, slope, group, pstats=None):
        assert training
        xd = x.to(ACC)
        mean, var = _stats(xd)
        rstd = torch.rsqrt(var + eps)
        z = (xd - mean) * rstd * (_shape(weight).abs() + eps) + _shape(bias)
        y = F.leaky_relu(z, slope) if activation == "leaky_relu" else z
        inplace = x.is_contiguous(memory_format=torch.channels_last)
        if inplace:
            x.copy_(y.to(x.dtype))
            ctx.mark_dirty(x)
            out = x
        else:
            out = _cl(y.to(x.dtype))
        ctx.save_for_backward(out, weight, bias, rstd)
        ctx.cfg = (eps, activation, slope)
        return out

    @staticmethod
    def backward(ctx, dy):
        y, weight, bias, rstd = ctx.saved_tensors
        eps, activation, slope = ctx.cfg
        z, dz = y.to(ACC), dy.to(y.dtype).to(ACC)
        if activation == "leaky_relu":
            neg = torch.signbit(y)            # (-0 is a negative pre-activation whose slope-fold underflowed: fp16)
            z = torch.where(neg, z / slope, z)
            dz = torch.where(neg, dz * slope, dz)
        gamma = _shape(weight).abs() + eps
        xhat = (z - _shape(bias)) / gamma
        n = y.numel() // y.shape[1]
        s1, s2 = dz.sum((0, 2, 3), keepdim=True), (dz * xhat).sum((0, 2, 3), keepdim=True)
        dx = gamma * rstd * (dz - s1 / n - xhat * s2 / n)
        dw = (s2.flatten() * torch.sign(weight.to(ACC))).to(weight.dtype)
        return (_cl(dx.to(y.dtype)), dw, s1.flatten().to(bias.dtype)) + (None,) * 9


def _bn(xd, wd, bd, eps):
    return F.batch_norm(xd, None, None, wd.abs() + eps, bd, True, 0.0, eps)


class _RefAbnAddRelu:
    """res_net.py:62-79 tail: relu(norm_identity(x) + shortcut); the norm's output is a stored 16-bit tensor in the reference"""

    @staticmethod
    def forward(ctx, x, shortcut, weight, bias, running_mean, running_var, training, momentum, eps, group, pstats=None):
        assert training
        ctx.save_for_backward(x, shortcut, weight, bias)
        ctx.eps = eps
        z = _bn(x.to(ACC), weight.to(ACC), bias.to(ACC), eps).to(x.dtype).to(ACC) + shortcut.to(ACC)
        return _cl(torch.relu(z).to(x.dtype))

    @staticmethod
    def backward(ctx, g):
        x, shortcut, weight, bias = ctx.saved_tensors
        with torch.enable_grad():
            xd, sd = x.to(ACC).requires_grad_(True), shortcut.to(ACC).requires_grad_(True)
            wd, bd = weight.to(ACC).requires_grad_(True), bias.to(ACC).requires_grad_(True)
            y = torch.relu(_ste_round(_bn(xd, wd, bd, ctx.eps), x.dtype) + sd)
        dx, ds, dw, db = torch.autograd.grad(y, [xd, sd, wd, bd], g.to(x.dtype).to(ACC))
        return (_cl(dx.to(x.dtype)), _cl(ds.to(x.dtype)), dw.to(weight.dtype), db.to(bias.dtype)) + (None,) * 7


class _RefAbnPool:
    """BasicStem: max_pool2d(norm(x), 3, 2, 1) on the stored (rounded) activations"""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, activation, slope, group, pstats=None):
        assert training
        ctx.save_for_backward(x, weight, bias)
        ctx.cfg = (eps, activation, slope)
        y = _RefAbnPool._act(_bn(x.to(ACC), weight.to(ACC), bias.to(ACC), eps), activation, slope)
        return _cl(F.max_pool2d(y.to(x.dtype).to(ACC), 3, 2, 1).to(x.dtype))

    @staticmethod
    def _act(y, activation, slope):
        return F.leaky_relu(y, slope) if activation == "leaky_relu" else y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        eps, activation, slope = ctx.cfg
        with torch.enable_grad():
            xd = x.to(ACC).requires_grad_(True)
            wd, bd = weight.to(ACC).requires_grad_(True), bias.to(ACC).requires_grad_(True)
            y = F.max_pool2d(_ste_round(_RefAbnPool._act(_bn(xd, wd, bd, eps), activation, slope), x.dtype), 3, 2, 1)
        dx, dw, db = torch.autograd.grad(y, [xd, wd, bd], dy.to(x.dtype).to(ACC))
        return (_cl(dx.to(x.dtype)), dw.to(weight.dtype), db.to(bias.dtype)) + (None,) * 9


def _step(m, batch, scale):
    m.zero_grad(set_to_none=True)
    losses = m(batch)
    (sum(losses.values()) * scale).backward()
    torch.cuda.synchronize()
    return ({k: float(v.detach()) for k, v in losses.items()},
            {n: (p.grad.detach().double() / scale).cpu().flatten() for n, p in m.named_parameters() if p.grad is not None})


def _substitute(monkeypatch, ops, calls):
    for cls, ref, kind in ((ops._ConvFn, _RefConv, "conv"), (ops._ShortcutS2Fn, _RefShortcutS2, "conv"), (ops._IABNFn, _RefIABN, "norm"), (ops._AbnAddReluFn, _RefAbnAddRelu, "norm"),
                           (ops._AbnPoolFn, _RefAbnPool, "norm")):
        def counted(f, kind=kind):
            def g(*a, **k):
                calls[kind] += 1
                return f(*a, **k)
            return staticmethod(g)
        monkeypatch.setattr(cls, "forward", counted(ref.forward))
        monkeypatch.setattr(cls, "backward", staticmethod(ref.backward))


# bf16 at 64x96 was part of this list until round 4 and passed only on a 10x looser bound.  It is not a meaningful comparison: with a 64x96 input
# res5 is 2x3 pixels (12 samples per deep norm, 2 for the global-context norm), and the gradients that have passed through all of that --
# the backbone's conv weights, 90 % of the squared gradient norm, 64 % in backbone.stem.conv1.weight alone -- are rounding noise in EVERY
# evaluation: against the fp64 rounding-matched reference the product has cosine -0.03 (stem) / 0.11-0.26 (res2-res5) at 1.1x the norm,
# the fp32 twin 0.46 / 0.51-0.55 at 2.3-2.8x the norm (profiles/r04_composed_64x96_diagnosis.txt; MGN_COMPOSED_DIAG=1 prints the table).
# The same tensors agree to 0.43 / 0.66-0.69 (twin 0.55 / 0.53-0.58) at 192x640 in bf16 and to 0.99 in fp16 (8x smaller ulp): the
# disagreement scales with the rounding unit and the sample count of the norms, not with the path.  What the small shape can prove -- every
# fused node right in composition -- is asserted sharply by test_step_nodes_gpu.py (<= 0.7 ulp per node from its own recorded inputs).
@pytest.mark.parametrize("dtype,H,W", [(torch.bfloat16, 192, 640), (torch.bfloat16, 256, 512), (torch.float16, 192, 640)])
def test_composed_step_is_as_close_to_the_rounding_matched_reference_as_its_fp32_twin(dtype, H, W, monkeypatch):
    import sys

    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.modeling import ops
    from test_grad_parity_gpu import _check_vs_torch_bf16

    monkeypatch.setenv("MGNET_STREAMS", "0")
    cfg, m = small_model(with_depth=True, seed=3)
    _randomise(m)
    m = m.cuda().train()
    m.amp_dtype = dtype
    batch = synthetic_batch(2, H, W, "cuda", seed=5)
    scale = 1.0 if dtype == torch.bfloat16 else 4096.0
    calls = {"conv": 0, "norm": 0}
    hip_losses, hip = _step(m, batch, scale)                     # (a) the product path
    _substitute(monkeypatch, ops, calls)
    ref_losses, ref = _step(m, batch, scale)                     # (b) fp64 nodes, 16-bit rounding at the product's points
    assert calls["conv"] >= 70 and calls["norm"] >= 53, calls    # ... and the run really went through the substituted nodes
    monkeypatch.setattr(sys.modules[__name__], "ACC", torch.float32)
    # (c) the same nodes in fp32 arithmetic: the control.  It runs through torch's fp32 convolutions, which are NOT reproducible on this stack
    # (round 5, six runs of this test: the product's numbers identical to four digits every time, the control's median cosine 0.940 ...
    # 0.958, its loss_center 25.690 ... 25.696 against the reference's 25.690) -- so it is evaluated three times and each tensor / loss
    # is given its WORST result, like the torch-bf16 yardstick of test_grad_parity_gpu.py: the question is whether the product lies inside
    # the spread of fp32 evaluations, not whether it beats one lucky draw
    twins = [_step(m, batch, scale) for _ in range(3)]
    twin_losses, twin = twins[0]
    assert set(hip) == set(ref) == set(twin) and len(hip) > 200
    rows = _rows(hip, ref)
    runs = [{r[0]: r for r in _rows(t[1], ref)} for t in twins]
    rows_twin = [(n, min(b[n][1] for b in runs), max(b[n][2] for b in runs), runs[0][n][3]) for n in runs[0]]
    twin_dist = {k: max(abs(t[0][k] - v) for t in twins) for k, v in ref_losses.items()}
    med = lambda v: sorted(v)[len(v) // 2]
    sig = {r[0] for r in _significant(rows)}
    whole = lambda a: float((torch.cat([a[n] for n in sorted(ref)]) @ torch.cat([ref[n] for n in sorted(ref)])) /
                            (torch.cat([a[n] for n in sorted(ref)]).norm() * torch.cat([ref[n] for n in sorted(ref)]).norm()))
    print(f"\n[composed {str(dtype)[6:]} {H}x{W}] {len(sig)} tensors vs the fp64 rounding-matched reference: "
          f"HIP median cosine {med([r[1] for r in rows if r[0] in sig]):.4f} / relative error {med([r[2] for r in rows if r[0] in sig]):.3f} / whole gradient {whole(hip):.4f};  "
          f"fp32 twin {med([r[1] for r in rows_twin if r[0] in sig]):.4f} / {med([r[2] for r in rows_twin if r[0] in sig]):.3f} / {whole(twin):.4f}")
    print("   losses HIP / fp64 reference / fp32 twin:", {k: (round(hip_losses[k], 5), round(v, 5), round(twin_losses[k], 5)) for k, v in ref_losses.items()})
    if os.environ.get("MGN_COMPOSED_DIAG"):
        # which tensors carry the whole-gradient comparison: share of the reference gradient's squared norm, and each tensor's cosine
        tot = sum(float(ref[n].norm()) ** 2 for n in ref)
        big = sorted(ref, key=lambda n: -float(ref[n].norm()))[:10]
        cosn = lambda a, n: float((a[n] @ ref[n]) / (a[n].norm() * ref[n].norm() + 1e-300))
        for n in big:
            print(f"      {n:58s} share {float(ref[n].norm()) ** 2 / tot:6.3f}  numel {ref[n].numel():8d}  cos HIP {cosn(hip, n):7.4f}  twin {cosn(twin, n):7.4f}  "
                  f"|HIP|/|ref| {float(hip[n].norm() / ref[n].norm()):.3f}  |twin|/|ref| {float(twin[n].norm() / ref[n].norm()):.3f}")
    for k, v in ref_losses.items():
        assert hip_losses[k] == pytest.approx(v, rel=2e-2, abs=2e-4), (k, hip_losses[k], v)
        # the product's losses are as close to the reference's as the control's are (3x its distance + a floor of 1e-3 relative)
        assert abs(hip_losses[k] - v) <= 3 * twin_dist[k] + 1e-3 * abs(v) + 1e-5, (k, hip_losses[k], v, [t[0][k] for t in twins])
    # (strict: the yardstick here is deterministic torch fp32 arithmetic, not MIOpen's bf16 convolutions -- measured 0 worse, 0 far worse)
    _check_vs_torch_bf16(rows, rows_twin, f"composed {str(dtype)[6:]} {H}x{W} ", worse_frac=0.05, strict=True,
                         labels=("HIP", "fp32 twin", "the fp64 rounding-matched reference"))
