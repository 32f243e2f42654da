"""GPU: the small fused tails of the step (round 3) against their torch formulations -- prediction-head activations on channel-padded
predictor outputs (csrc/headact.hip; mg_net.py:694, :819-823), the uncertainty weighting (csrc/scalars.hip; mg_net.py:360-372), the
decoder's `arm(x) + last` inside the attention scaling pass (layers.py:87), PoseCNN's ReLU / bias / mean tail (layers.py:155-167)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kind,C", [("none", 20), ("sigmoid", 1), ("sigmoid2", 1), ("none", 2)])
def test_head_activation_on_padded_predictor_output(kind, C, dtype):
    from mgnet_amd.modeling import ops
    torch.manual_seed(0)
    B, h, w = 2, 9, 14
    xp = (torch.randn(B, 32, h, w, device="cuda") * 2).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = ops.head_activation(ops.PaddedMap(xp, C), kind)
    assert y.dtype == torch.float32 and y.shape == (B, C, h, w) and y.is_contiguous()
    xr = xp.detach()[:, :C].double().requires_grad_(True)
    yr = xr if kind == "none" else (torch.sigmoid(xr) if kind == "sigmoid" else torch.sigmoid(xr) / 0.5)
    assert float((y.double() - yr).abs().max()) < 2e-6
    g = torch.randn(B, C, h, w, device="cuda")
    y.backward(g)
    yr.backward(g.double())
    got = xp.grad
    assert got.shape == xp.shape and got.dtype == dtype and got.is_contiguous(memory_format=torch.channels_last)
    assert float(got[:, C:].abs().max()) == 0.0                     # padding channels: exact zeros
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert float((got[:, :C].double() - xr.grad).abs().max()) <= ulp * float(xr.grad.abs().max()) + 1e-7
    # strided gradient (the NHWC tables of the loss kernels)
    tab = torch.randn(B, h, w, 4, device="cuda")
    xp2 = xp.detach().clone().requires_grad_(True)
    y2 = ops.head_activation(ops.PaddedMap(xp2, 1), "sigmoid")
    y2.backward(tab[..., 2].unsqueeze(1))
    ref = (tab[..., 2].unsqueeze(1).double() * (y2.double() * (1 - y2.double()))).detach()
    assert float((xp2.grad[:, :1].double() - ref).abs().max()) <= ulp * float(ref.abs().max()) + 1e-7


def test_uncertainty_weighting_matches_reference_formula():
    from mgnet_amd.modeling import ops
    torch.manual_seed(1)
    keys = ["loss_sem_seg", "loss_center", "loss_offset", "loss_photometric", "loss_smoothness"]
    for n in (5, 3):
        raw0 = [torch.rand((), device="cuda") * 3 for _ in range(n)]
        lv0 = torch.randn(5, device="cuda") * 0.4
        raws = [r.clone().requires_grad_(True) for r in raw0]
        lv = lv0.clone().requires_grad_(True)
        weighted, raw, unc = ops.uncertainty_weighting({k: r for k, r in zip(keys, raws)}, lv)
        coef = torch.tensor([1.7, 0.3, 1.0, 2.0, 0.5], device="cuda")[:n]
        (torch.stack([weighted[k] for k in keys[:n]]) * coef).sum().backward()
        r2 = [r.clone().double().requires_grad_(True) for r in raw0]
        l2 = lv0.clone().double().requires_grad_(True)
        want = []
        for i, k in enumerate(keys[:n]):   # mg_net.py:362-371
            tau = 1.0 if k == "loss_sem_seg" else 0.5
            want.append(tau * torch.exp(-l2[i]) * r2[i] + 0.5 * l2[i])
        (torch.stack(want) * coef.double()).sum().backward()
        for i, k in enumerate(keys[:n]):
            assert float(weighted[k]) == pytest.approx(float(want[i]), rel=2e-6)
            assert float(raw[k]) == float(raw0[i]) and float(unc[k]) == pytest.approx(float(torch.exp(lv0[i])), rel=2e-6)
            assert float(raws[i].grad) == pytest.approx(float(r2[i].grad), rel=2e-6)
        assert torch.allclose(lv.grad.double(), l2.grad, rtol=2e-6, atol=1e-7) and float(lv.grad[n:].abs().sum()) == 0.0
    # per head (MGNet.forward: the tasks of one head per call, k0 = its first task): bit-identical values and gradients to the one call
    # over the whole dictionary, started from the task losses themselves (Trainer._backward) as from their weighted sum
    raw0 = [torch.rand((), device="cuda") * 3 for _ in range(5)]
    ra, rb = [r.clone().requires_grad_(True) for r in raw0], [r.clone().requires_grad_(True) for r in raw0]
    la, lb = lv0.clone().requires_grad_(True), lv0.clone().requires_grad_(True)
    wa, _, ua = ops.uncertainty_weighting(dict(zip(keys, ra)), la)
    sum(wa.values()).backward()
    wb, ub = {}, {}
    for k0, n in ((3, 2), (1, 2), (0, 1)):   # (the depth head's tasks first, as the forward issues them)
        w, _, u = ops.uncertainty_weighting({k: r for k, r in zip(keys[k0:k0 + n], rb[k0:k0 + n])}, lb, k0=k0)
        wb.update(w)
        ub.update(u)
    one = torch.ones((), device="cuda")
    torch.autograd.backward([wb[k] for k in keys], grad_tensors=[one] * 5)
    for i, k in enumerate(keys):
        assert float(wa[k]) == float(wb[k]) and float(ua[k]) == float(ub[k]) and float(ra[i].grad) == float(rb[i].grad), k
    assert torch.equal(la.grad, lb.grad)
    # an unused output gets no gradient; the others are unaffected
    raws = [r.clone().requires_grad_(True) for r in raw0[:3]]
    lv = lv0.clone().requires_grad_(True)
    weighted, _, _ = ops.uncertainty_weighting({k: r for k, r in zip(keys, raws)}, lv)
    weighted["loss_center"].backward()
    assert float(raws[0].grad) == 0.0 and float(raws[2].grad) == 0.0 and float(raws[1].grad) == pytest.approx(0.5 * float(torch.exp(-lv0[1])), rel=2e-6)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_arm_plus_last_in_one_pass(dtype, monkeypatch):
    """AttentionRefinementModule(x) + last with the addend folded into the scaling kernel: same values as the separate add up to ONE
    rounding instead of two, gradients of x, the parameters and `last` vs the unfused path"""
    from mgnet_amd.modeling.layers import AttentionRefinementModule
    torch.manual_seed(2)
    arm = AttentionRefinementModule(64, 128).cuda().train()
    x0 = torch.randn(2, 64, 12, 20, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    last0 = torch.randn(2, 128, 12, 20, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    g = torch.randn(2, 128, 12, 20, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    res = []
    for fused in (True, False):
        for p in arm.parameters():
            p.grad = None
        for b in arm.buffers():
            b.zero_() if "mean" in str(b.shape) else None
        x, last = x0.clone().requires_grad_(True), last0.clone().requires_grad_(True)
        y = arm(x, last) if fused else arm(x) + last
        y.backward(g)
        res.append((y.detach().float(), x.grad.float(), last.grad.float(), [p.grad.clone().float() for p in arm.parameters()]))
    (y1, dx1, dl1, dp1), (y2, dx2, dl2, dp2) = res
    spacing = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10     # upper bound of the 16-bit grid spacing relative to |y|
    assert float((y1 - y2).abs().max()) <= 1.01 * spacing * float(y2.abs().max())   # one grid step at the largest value
    assert torch.equal(dl1, dl2) and torch.equal(dl1, g.float())
    assert torch.equal(dx1, dx2) and all(torch.equal(a, b) for a, b in zip(dp1, dp2))


def test_posecnn_tail_matches_torch(monkeypatch):
    """relu_(conv + bias) with the ReLU in the epilogue and its mask in the backward, bias gradient by the column-sum kernel, and
    0.01 * mean over H, W of the 12-channel (padded to 32) pose map with its broadcast adjoint, against fp64 torch"""
    from mgnet_amd.modeling import ops
    torch.manual_seed(3)
    x0 = torch.randn(2, 256, 6, 10, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w1 = torch.nn.Parameter(torch.randn(256, 256, 3, 3, device="cuda") * 0.03)
    b1 = torch.nn.Parameter(torch.randn(256, device="cuda") * 0.1)
    w4 = torch.nn.Parameter(torch.randn(12, 256, 1, 1, device="cuda") * 0.05)
    b4 = torch.nn.Parameter(torch.randn(12, device="cuda") * 0.1)
    x = x0.clone().requires_grad_(True)
    h = ops.conv2d(x, w1, b1, 1, 1, relu=True)
    out = ops.mean_hw(ops.conv2d(h, w4, b4, 1, 0, keep_pad=True), 0.01)
    assert out.shape == (2, 12) and out.dtype == torch.float32
    gq = torch.randn(2, 12, device="cuda")
    out.backward(gq)
    xd = x0.double().requires_grad_(True)
    p = [t.detach().double().requires_grad_(True) for t in (w1, b1, w4, b4)]
    q = lambda t: t.bfloat16().double()
    hd = torch.relu(F.conv2d(xd, q(p[0].detach()) + (p[0] - p[0].detach()), p[1], padding=1))
    hq = hd + (q(hd.detach()) - hd.detach())                  # the stored activation is rounded to bf16
    od = 0.01 * (F.conv2d(hq, q(p[2].detach()) + (p[2] - p[2].detach()), p[3])).mean((2, 3))
    od.backward(gq.double())
    rel = lambda a, b: float((a.double() - b).norm() / (b.norm() + 1e-30))
    assert rel(out, od) < 3e-3
    for name, got, ref in (("dx", x.grad, xd.grad), ("dw1", w1.grad, p[0].grad), ("db1", b1.grad, p[1].grad), ("dw4", w4.grad, p[2].grad),
                           ("db4", b4.grad, p[3].grad)):
        assert rel(got, ref) < 1.5e-2, (name, rel(got, ref))


@pytest.mark.parametrize("kind,dtype", [("arm", torch.bfloat16), ("ffm", torch.bfloat16), ("arm", torch.float16), ("ffm", torch.float16)])
def test_norm_and_attention_fused_node_vs_separate_nodes_and_fp64(kind, dtype, monkeypatch):
    """ops._AbnAttentionFn (conv output -> InPlaceABNSync -> ARM / FFM attention with the passes over the map fused: pool inside the norm's
    apply pass; in the backward five per-(image, channel) sums of ONE pass over (g, z) feed both the attention branch and the norm's sums)
    against (a) the two separate nodes it replaces (MGN_NO_ABN_ATTN=1) and (b) the oracle's fp64 evaluation of layers.py:221-322 on the
    same 16-bit inputs: at least as close to fp64 as the separate nodes (it rounds the intermediate gradient once less)"""
    import oracle.network_oracle as NO
    from mgnet_amd.modeling.layers import AttentionRefinementModule, FeatureFusionModule
    torch.manual_seed(5)
    N, H, W = 3, 20, 36
    if kind == "arm":
        mod, prefix = AttentionRefinementModule(64, 128).cuda().train(), "m"
        ins0 = [torch.randn(N, 64, H, W, device="cuda")]
    else:
        mod, prefix = FeatureFusionModule(128 + 128, 256).cuda().train(), "m"
        ins0 = [torch.randn(N, 128, H, W, device="cuda"), torch.randn(N, 128, H, W, device="cuda")]
    with torch.no_grad():
        for n_, p in mod.named_parameters():
            if n_.endswith("norm.weight"):
                p.copy_(torch.randn_like(p) * 0.5 + 1.0)      # (signed: the norm uses |weight| + eps and d weight carries the sign)
            elif n_.endswith("norm.bias"):
                p.copy_(torch.randn_like(p) * 0.3)
    ins0 = [t.to(dtype).contiguous(memory_format=torch.channels_last) for t in ins0]
    Cout = 128 if kind == "arm" else 256
    g = torch.randn(N, Cout, H, W, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    res = {}
    for tag in ("fused", "separate"):
        if tag == "separate":
            monkeypatch.setenv("MGN_NO_ABN_ATTN", "1")
        for p in mod.parameters():
            p.grad = None
        ins = [t.clone().requires_grad_(True) for t in ins0]
        y = mod(*ins)
        y.backward(g)
        res[tag] = (y.detach().double(), [t.grad.double() for t in ins], {n_: p.grad.double().clone() for n_, p in mod.named_parameters()})
    monkeypatch.delenv("MGN_NO_ABN_ATTN")
    # fp64 oracle on the same (16-bit rounded) inputs
    sd = {f"{prefix}.{k}": v.detach().double().clone().requires_grad_(v.dtype.is_floating_point) for k, v in mod.state_dict().items()}
    ins = [t.double().clone().requires_grad_(True) for t in ins0]
    yr = NO.arm(sd, prefix, ins[0]) if kind == "arm" else NO.ffm(sd, prefix, ins[0], ins[1])
    yr.backward(g.double())
    spacing = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    yf, ys = res["fused"][0], res["separate"][0]
    assert float((yf - yr).abs().max()) <= 3 * spacing * float(yr.abs().max())
    assert float((yf - ys).abs().max()) <= 2.02 * spacing * float(yr.abs().max())

    def rel(a, b):
        return float((a - b).norm() / b.norm().clamp_min(1e-30))
    rows = [("input%d" % i, res["fused"][1][i], res["separate"][1][i], ins[i].grad) for i in range(len(ins))]
    rows += [(n_, res["fused"][2][n_], res["separate"][2][n_], sd[f"{prefix}.{n_}"].grad) for n_ in res["fused"][2]]
    for name, gf, gs, gr in rows:
        ef, es = rel(gf, gr), rel(gs, gr)
        tol = 4e-2 if dtype == torch.bfloat16 else 2e-2   # (the conv's own 16-bit weights and gradients bound both paths: 1.05e-2 in fp16)
        assert ef <= tol, (kind, name, ef, es)
        assert ef <= 1.25 * es + 1e-3, (kind, name, ef, es)
