"""GPU: fused upsample+loss kernels (mgnet_amd/csrc/headloss.hip) against the reference formulation evaluated with
torch ops in fp32 on the same bf16-rounded low-res maps (F.interpolate + loss.py:45-81 OhemCE with the full sort /
mg_net.py:697-715), and against the golden OhemCE/DeepLabCE vectors generated from the reference."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _padded_lr(B, K, h, w, seed):
    """low-res head output as the conv kernel produces it: channels-last bf16, channels padded to 32, sliced to K"""
    g = torch.Generator().manual_seed(seed)
    full = torch.zeros(B, 32, h, w).contiguous(memory_format=torch.channels_last)
    full[:, :K] = torch.randn(B, K, h, w, generator=g) * 2
    return full.to(torch.bfloat16).cuda()[:, :K]


def _ohem_ref(logits, labels, weights, ignore, thr, n_min):  # loss.py:67-81, with the sort
    pl = (F.cross_entropy(logits, labels, ignore_index=ignore, reduction="none") * weights).contiguous().view(-1)
    pl, _ = torch.sort(pl, descending=True)
    t = -torch.log(torch.tensor(thr, dtype=torch.float))
    pl = pl[pl > t] if pl[n_min] > t else pl[:n_min]
    return pl.mean()


@pytest.mark.parametrize("cfg", [(2, 20, 6, 10, 8, 0.7, 500), (1, 19, 5, 7, 8, 0.05, 1600), (2, 7, 9, 9, 8, 0.7, 100000 // 40),
                                 (1, 20, 4, 6, 16, 0.999, 1000)])
def test_fused_ohem_matches_sorted_reference(cfg):
    from mgnet_amd.modeling import ops

    B, K, h, w, s, thr, n_min = cfg
    H, W = h * s, w * s
    lr = _padded_lr(B, K, h, w, seed=sum(cfg[:5])).requires_grad_(True)
    g = torch.Generator().manual_seed(1)
    labels = torch.randint(0, K, (B, H, W), generator=g)
    labels[torch.rand(B, H, W, generator=g) < 0.1] = 255
    weights = torch.where(torch.rand(B, H, W, generator=g) < 0.2, 3.0, 1.0)
    # reference: materialised fp32 upsampling + sort-based OHEM
    lr_ref = lr.detach().float().cpu().requires_grad_(True)
    full = F.interpolate(lr_ref, scale_factor=s, mode="bilinear", align_corners=True)
    ref = _ohem_ref(full, labels, weights, 255, thr, n_min)
    ref.backward()
    got = ops.upsampled_ce(ops.LazyUpsample(lr, s), labels.cuda(), weights.cuda(), 255, "ohem", float(-np.log(thr)), n_min)
    (got * 1.7).backward()
    assert float(got) == pytest.approx(float(ref), rel=2e-5), (float(got), float(ref))
    gr = lr_ref.grad * 1.7
    err = (lr.grad.float().cpu() - gr).abs().max() / gr.abs().max()
    assert float(err) < 1e-2, float(err)  # gradient is rounded to bf16 (2^-8) on return


def test_golden_ohem_and_deeplab_vectors():
    """scale 1 (H == h): the fused kernels must reproduce the reference's own OhemCE / DeepLabCE values (bf16 logits)."""
    from mgnet_amd.modeling.loss import DeepLabCE, OhemCE
    from mgnet_amd.modeling import ops

    z = np.load(os.path.join(GOLDEN, "ce_losses.npz"))
    logits = torch.from_numpy(z["logits"])
    B, K, H, W = logits.shape
    full = torch.zeros(B, 32, H, W).contiguous(memory_format=torch.channels_last)
    full[:, :K] = logits
    lr_full = full.to(torch.bfloat16).cuda()
    lr = lr_full[:, :K]
    labels, weights = torch.from_numpy(z["labels"]).cuda(), torch.from_numpy(z["weights"]).cuda()
    for tag in ("ohem_top", "ohem_thr", "ohem_top_hi"):
        n_min, thr = int(z[tag + "_cfg"][0]), float(z[tag + "_cfg"][1])
        crit = OhemCE(ignore_label=255, ohem_threshold=thr, n_min=n_min)
        xf = lr_full.detach().clone().requires_grad_(True)   # padded channels-last storage, like the conv output
        v = crit(ops.LazyUpsample(xf[:, :K], 1), labels, weights)
        assert float(v) == pytest.approx(float(z[tag + "_val"]), rel=1.5e-2), tag      # bf16-rounded logits
        # (the backward tile kernel supports upsampling factors >= 8 only; gradients are pinned by the sorted-reference test)
    for tag in ("dl_all", "dl_top"):
        k = float(z[tag + "_cfg"][0])
        xf = lr_full.detach().clone().requires_grad_(True)
        v = DeepLabCE(ignore_label=255, top_k_percent_pixels=k)(ops.LazyUpsample(xf[:, :K], 1), labels, weights)
        assert float(v) == pytest.approx(float(z[tag + "_val"]), rel=1.5e-2), tag
    with pytest.raises(IndexError):  # SURVEY section 4 KAT: n_min >= pixel count
        OhemCE(ignore_label=255, n_min=B * H * W)(ops.LazyUpsample(lr, 1), labels, weights)


@pytest.mark.parametrize("shape", [(2, 6, 10), (1, 5, 9)])
def test_fused_instance_losses(shape):
    from mgnet_amd.modeling import ops

    B, h, w = shape
    s = 8
    H, W = h * s, w * s
    g = torch.Generator().manual_seed(B + h)
    center_lr = torch.sigmoid(torch.randn(B, 1, h, w, generator=g)).cuda().requires_grad_(True)
    offset_lr = _padded_lr(B, 2, h, w, seed=3).requires_grad_(True)
    ct = torch.rand(B, 1, H, W, generator=g)
    cw = (torch.rand(B, 1, H, W, generator=g) < 0.7).float()
    ot = torch.rand(B, 2, H, W, generator=g) * 128 - 64
    ow = (torch.rand(B, 1, H, W, generator=g) < 0.3).float()
    c_ref = center_lr.detach().cpu().requires_grad_(True)
    o_ref = offset_lr.detach().float().cpu().requires_grad_(True)
    cu = F.interpolate(c_ref, scale_factor=s, mode="bilinear", align_corners=True)
    ou = F.interpolate(o_ref, scale_factor=s, mode="bilinear", align_corners=True) * s
    lc = ((cu - ct) ** 2 * cw).sum() / cw.sum()          # mg_net.py:697-705
    lo = ((ou - ot).abs() * ow).sum() / ow.sum()         # mg_net.py:706-711
    (2.0 * lc + 0.3 * lo).backward()
    tg = {"center": ct.cuda(), "center_weights": cw.cuda(), "offset": ot.cuda(), "offset_weights": ow.cuda()}
    l2 = ops.upsampled_ins_losses(ops.LazyUpsample(center_lr, s), ops.LazyUpsample(offset_lr, s, mult=float(s)), tg)
    (2.0 * l2[0] + 0.3 * l2[1]).backward()
    assert float(l2[0]) == pytest.approx(float(lc), rel=2e-5) and float(l2[1]) == pytest.approx(float(lo), rel=2e-5)
    assert float((center_lr.grad.cpu() - c_ref.grad).abs().max() / c_ref.grad.abs().max()) < 1e-4
    assert float((offset_lr.grad.float().cpu() - o_ref.grad).abs().max() / o_ref.grad.abs().max()) < 1e-2
    # all-zero weights: loss 0 and zero gradients (the `.sum() > 0` branches of the reference)
    tg0 = dict(tg, center_weights=torch.zeros_like(tg["center_weights"]), offset_weights=torch.zeros_like(tg["offset_weights"]))
    c2 = center_lr.detach().clone().requires_grad_(True)
    z2 = ops.upsampled_ins_losses(ops.LazyUpsample(c2, s), ops.LazyUpsample(offset_lr.detach(), s, mult=float(s)), tg0)
    z2.sum().backward()
    assert float(z2.abs().sum()) == 0.0 and float(c2.grad.abs().max()) == 0.0


@pytest.mark.parametrize("cfg", [(2, 6, 10, 8), (1, 5, 7, 16), (2, 3, 4, 32)])
def test_depth_upsample_fwd_bwd(cfg):
    from mgnet_amd.modeling import ops

    B, h, w, s = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x0 = torch.rand(B, 1, h, w, generator=g) * 2
    xr = x0.clone().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=s, mode="bilinear", align_corners=True)
    gy = torch.randn(*yr.shape, generator=g)
    (yr * gy).sum().backward()
    x = x0.cuda().requires_grad_(True)
    y = ops.upsample_bilinear(x, s)
    (y * gy.cuda()).sum().backward()
    assert torch.allclose(y.cpu(), yr, atol=1e-6, rtol=1e-6)
    assert torch.allclose(x.grad.cpu(), xr.grad, atol=1e-4, rtol=1e-5)


@pytest.mark.parametrize("cfg", [(2, 19, 16, 24, 8), (1, 7, 9, 13, 8), (3, 32, 5, 6, 16), (2, 19, 33, 17, 8)])
def test_bilinear_adjoint_is_reproducible_and_equals_the_atomic_form(cfg, monkeypatch):
    """The tile-wise bilinear adjoints (semantic / instance / depth heads) in their default form -- every tile stores its footprint,
    a gather sums them in a fixed order into an UNINITIALISED destination -- give the same bits on every run and agree with the
    float-atomic form (MGN_ADJOINT_ATOMICS=1) to summation-order noise, on ragged geometries (partial tiles, tiny low-res maps)."""
    from mgnet_amd import _C

    B, K, h, w, sc = cfg
    H, W = h * sc - 3, w * sc - 5            # not a multiple of the 32 x 16 tile, align_corners geometry
    torch.manual_seed(sum(cfg))
    Kc = (K + 7) // 8 * 8
    lr = torch.randn(B, Kc, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)[:, :K]
    labels = torch.randint(0, K, (B, H, W), device="cuda")
    labels[torch.rand(B, H, W, device="cuda") < 0.1] = 255
    weights = torch.rand(B, H, W, device="cuda") + 0.5
    ce, sums = _C.upce_fwd(lr, labels, weights, H, W, 255, 0.3)
    sel, _ = _C.ohem_select(ce, sums, 0.3, max(1, B * H * W // 16), False)
    sel = sel.float().contiguous()
    g = torch.ones(1, device="cuda")
    dfull = torch.randn(B, 1, H, W, device="cuda")

    def run():
        torch.empty(64 << 20, device="cuda").fill_(float("nan"))   # poison the allocator's free blocks: nothing may rely on zeros
        return _C.upce_bwd(lr, labels, weights, H, W, 255, ce, sel, g, Kc).clone(), _C.upsample1_bwd(dfull, h, w).clone()

    a1, b1 = run()
    a2, b2 = run()
    assert torch.equal(a1, a2) and torch.equal(b1, b2)
    assert torch.isfinite(a1).all() and torch.isfinite(b1).all() and float(a1[..., K:].abs().max() if Kc > K else 0.0) == 0.0
    monkeypatch.setenv("MGN_ADJOINT_ATOMICS", "1")
    a3, b3 = run()
    assert float((a1 - a3).abs().max()) <= 2e-6 * float(a3.abs().max()) + 1e-12
    assert float((b1 - b3).abs().max()) <= 2e-6 * float(b3.abs().max()) + 1e-12
