"""CPU: pin the C oracle (oracle/reproj_oracle.c) against vectors produced by the reference itself."""
import numpy as np
import pytest

import oracle
from conftest import REPROJ_CASES, golden_case_inputs, load_golden


def test_kats_pose_and_intrinsics():
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "kats.npz"))
    R, t = oracle.pose_vec2mat(np.concatenate([np.zeros((3, 3), np.float32), z["euler_in"]], 1))
    np.testing.assert_allclose(R, z["euler_out"], atol=2e-7)
    R, t = oracle.pose_vec2mat(z["vec_in"])
    np.testing.assert_allclose(R[0], z["vec_mat"][0, :3, :3], atol=2e-7)
    np.testing.assert_allclose(t[0], z["vec_mat"][0, :3, 3], atol=0)
    # euler2mat(0) = I exactly (SURVEY section 4 KAT)
    R0, _ = oracle.pose_vec2mat(np.zeros((1, 6), np.float32))
    assert np.array_equal(R0[0], np.eye(3, dtype=np.float32))
    np.testing.assert_allclose(oracle.kinv(z["scale_K_in"]), z["Kinv"], rtol=1e-7)
    np.testing.assert_array_equal(oracle.inv2depth(z["inv2depth_in"]), z["inv2depth_out"])


def test_kats_ssim():
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "kats.npz"))
    rs = np.random.RandomState(0)
    x = rs.uniform(0, 1, (1, 3, 6, 7)).astype(np.float32)
    # ssim(x, x) == 0 up to fp32 rounding of (E[x^2]-mu^2) (the reference's own value is ~1e-5, not 0)
    assert np.all(oracle.ssim(x, x) <= np.maximum(z["ssim_xx"].max(), 1e-4))
    assert np.all(oracle.ssim(x, x, prec="f64") < 1e-9)
    got = oracle.ssim(np.zeros((1, 1, 5, 5), np.float32), np.ones((1, 1, 5, 5), np.float32))
    np.testing.assert_allclose(got, z["ssim_01"], rtol=1e-6)
    assert abs(float(got.flat[0]) - 0.49995) < 1e-6


@pytest.mark.parametrize("name", REPROJ_CASES)
def test_stages_match_reference(name):
    c = golden_case_inputs(name)
    _, out = load_golden("reproj_" + name)
    K = c["K"][:, :3, :3]
    W = c["img"].shape[-1]
    ctx = [c["prev"], c["nxt"]]
    for j in range(2):
        R, t = oracle.pose_vec2mat(c["poses"][:, j])
        np.testing.assert_allclose(R, out[f"pose_mat{j}"][:, :3, :3], atol=3e-7)
        un = oracle.photometric(ctx[j], c["img"])
        np.testing.assert_allclose(un, out[f"unwarped{j}"], atol=2e-5)
        for i in range(3):
            w = oracle.view_synthesis(ctx[j], c["inv"][i], K, c["poses"][:, j])
            ref_w = out[f"warped{j}_{i}"]
            # coordinate round-off differs between torch's bmm and our scalar code: compare with the
            # stated tolerance on the bulk and allow isolated floor()-boundary flips on noise images
            err = np.abs(w - ref_w)
            tol = 1e-5 * max(1.0, W / 64.0)
            frac_bad = float((err > 50 * tol).mean())
            assert frac_bad < 2e-3, (name, j, i, frac_bad, err.max())
            ph = oracle.photometric(ref_w, c["img"])  # photometric stage on the reference's own warped image
            np.testing.assert_allclose(ph, out[f"photo{j}_{i}"], atol=2e-5)
    for i in range(3):
        sx, sy = oracle.calc_smoothness(c["inv"][i], c["img"])
        np.testing.assert_allclose(sx, out[f"smooth_x{i}"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(sy, out[f"smooth_y{i}"], rtol=2e-5, atol=2e-6)


def _check_full(name, c, out, dense=True):
    r = oracle.reproj_loss(c["inv"], c["img"], c["prev"], c["nxt"], c["mask"], c["K"], c["poses"],
                           g_photo=1.0, g_smooth=0.0, want_minmap=dense)
    r64 = oracle.reproj_loss(c["inv"], c["img"], c["prev"], c["nxt"], c["mask"], c["K"], c["poses"],
                             g_photo=1.0, g_smooth=0.0, prec="f64")
    rs_ = oracle.reproj_loss(c["inv"], c["img"], c["prev"], c["nxt"], c["mask"], c["K"], c["poses"],
                             g_photo=0.0, g_smooth=1.0)
    lp, ls = float(out["loss_photometric"]), float(out["loss_smoothness"])
    assert abs(r["loss_photometric"] - lp) <= 2e-5 * max(abs(lp), 1e-3), (r["loss_photometric"], lp)
    assert abs(r["loss_smoothness"] - ls) <= 2e-5 * abs(ls), (r["loss_smoothness"], ls)
    return r, r64, rs_


def _grad_close(got, ref, truth=None, rtol=2e-3, name=""):
    """Gradients through floor()/min()/sign() are piecewise: a 1-ulp coordinate difference can flip a
    bilinear cell or an arg-min at isolated pixels.  Require the bulk to agree tightly and the
    outliers to be rare; scale is the reference's own magnitude."""
    scale = np.abs(ref).max() + 1e-30
    if scale < 1e-10:  # degenerate case (identity pose + identical frames): the gradient is round-off noise
        assert np.abs(got).max() < 1e-9, name
        return
    err = np.abs(got - ref) / scale
    frac = float((err > rtol).mean())
    assert frac < 5e-3, (name, "fraction of mismatching grad elements", frac, float(err.max()))
    # aggregate agreement
    denom = np.abs(ref).sum() + 1e-30
    assert np.abs(got - ref).sum() / denom < 2e-2, (name, np.abs(got - ref).sum() / denom)


@pytest.mark.parametrize("name", REPROJ_CASES)
def test_full_loss_and_grads_match_reference(name):
    c = golden_case_inputs(name)
    _, out = load_golden("reproj_" + name)
    r, r64, rs_ = _check_full(name, c, out)
    for i in range(3):
        # SSIM's sigma = E[x^2]-mu^2 cancels catastrophically on smooth images, so single pixels may differ by
        # a few 1e-5 between two fp32 evaluation orders: tight on the bulk, loose cap on the tail
        err = np.abs(r["minmap"][i] - out[f"minmap{i}"])
        assert float((err > 3e-5).mean()) < 2e-3 and err.max() < 5e-4, (name, i, err.max())
        _grad_close(r["d_inv"][i], out[f"dphot_dinv{i}"], name=f"{name}/dphot_dinv{i}")
        ref_s = out[f"dsmooth_dinv{i}"]
        np.testing.assert_allclose(rs_["d_inv"][i], ref_s, rtol=1e-3, atol=1e-4 * np.abs(ref_s).max())
    ref_p = out["dphot_dposes"]
    if name == "identity_pose":
        # warped_next == image up to round-off, so the arg-min between warp_next/unwarp_next and the SSIM
        # gradient are round-off noise in the reference too: only require "numerically zero"
        assert np.abs(r["d_poses"]).max() < 1e-5 and np.abs(ref_p).max() < 1e-5
    else:
        np.testing.assert_allclose(r["d_poses"], ref_p, rtol=5e-3, atol=5e-3 * np.abs(ref_p).max() + 1e-9)
    assert np.all(rs_["d_poses"] == 0)


def test_survey_case_scalars():
    """B=2 192x640 (SURVEY Appendix E shape): inputs regenerated from the seed, outputs from the fixture."""
    c = golden_case_inputs("survey_192x640")
    _, out = load_golden("reproj_survey_192x640")
    r, r64, rs_ = _check_full("survey", c, out, dense=False)
    ref_p = out["dphot_dposes"]
    np.testing.assert_allclose(r["d_poses"], ref_p, rtol=2e-2, atol=2e-2 * np.abs(ref_p).max())
    for i in range(3):
        _grad_close(r["d_inv"][i][:, :, ::16, ::16], out[f"dphot_dinv{i}_s"], name=f"survey/dphot{i}")
        got_abs = np.abs(r["d_inv"][i].astype(np.float64)).sum()
        assert abs(got_abs - float(out[f"dphot_dinv{i}_abssum"])) < 1e-2 * float(out[f"dphot_dinv{i}_abssum"])
        ref_s = out[f"dsmooth_dinv{i}_s"]
        np.testing.assert_allclose(rs_["d_inv"][i][:, :, ::16, ::16], ref_s, rtol=1e-3, atol=1e-4 * np.abs(ref_s).max())


OPTION_COMBOS = ((False, "min", "zeros"), (False, "mean", "zeros"), (True, "min", "border"), (True, "min", "reflection"), (False, "mean", "border"))
# ("oob_clamp" + "reflection" is left out: that case has points behind the camera whose projected positions are ~1e12 pixels; folding such a
#  value back into the image is chaotic in fp32 -- one ulp of the position, i.e. the order of two multiplications upstream, moves the result
#  across the whole image -- so the reference's own value there is round-off noise (its CPU and GPU kernels fold differently, too))
OPTION_CASES = [(n, a, r, p) for n in ("rand_small", "oob_clamp", "no_mask_odd") for a, r, p in OPTION_COMBOS if not (n == "oob_clamp" and p == "reflection")]


@pytest.mark.parametrize("name,automask,reduce_op,padding_mode", OPTION_CASES)
def test_non_default_options_match_reference(name, automask, reduce_op, padding_mode):
    """automask_loss=False with photometric_reduce_op "min" / "mean" (loss.py:92-109, 131-144, 242-246) and padding_mode "border" /
    "reflection" of the warp (camera_utils.py:24-55) against the reference's own outputs (tests/golden/reproj_options.npz,
    make_golden_options.py)"""
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "reproj_options.npz"))
    tag = f"{name}.{int(automask)}.{reduce_op}" + ("" if padding_mode == "zeros" else "." + padding_mode)
    key = lambda k: z[f"{tag}.{k}"]
    c = golden_case_inputs(name)
    r = oracle.reproj_loss(c["inv"], c["img"], c["prev"], c["nxt"], c["mask"], c["K"], c["poses"], g_photo=1.0, g_smooth=0.0,
                           automask=automask, reduce_op=reduce_op, padding_mode=padding_mode)
    assert float(r["loss_photometric"]) == pytest.approx(float(key("loss_photometric")), rel=2e-5)
    assert float(r["loss_smoothness"]) == pytest.approx(float(key("loss_smoothness")), rel=2e-5)
    for i in range(3):
        _grad_close(r["d_inv"][i], key(f"dphot_dinv{i}"), name=f"{tag}/dphot_dinv{i}")
    ref_p = key("dphot_dposes")
    np.testing.assert_allclose(r["d_poses"], ref_p, rtol=5e-3, atol=5e-3 * np.abs(ref_p).max() + 1e-9)
    with pytest.raises(ValueError):
        oracle.reproj_loss(c["inv"], c["img"], c["prev"], c["nxt"], c["mask"], c["K"], c["poses"], automask=True, reduce_op="mean")


SSIM0_COMBOS = ((True, "min", "zeros"), (False, "min", "zeros"), (False, "mean", "zeros"), (True, "min", "border"))


@pytest.mark.parametrize("name", ["rand_small", "oob_clamp", "no_mask_odd"])
@pytest.mark.parametrize("automask,reduce_op,padding_mode", SSIM0_COMBOS)
def test_ssim_weight_zero_matches_reference(name, automask, reduce_op, padding_mode):
    """ssim_loss_weight = 0 (loss.py:185,196-197): the photometric maps are the 3-channel L1 maps; "min" runs over channels and sources
    and needs a reprojection mask, "mean" only exists without one -- the other two combinations raise IndexError in the reference
    (`<tag>.raises` in the fixture) and here."""
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "reproj_options.npz"))
    tag = f"{name}.{int(automask)}.{reduce_op}" + ("" if padding_mode == "zeros" else "." + padding_mode) + ".ssim0"
    c = golden_case_inputs(name)
    call = lambda: oracle.reproj_loss(c["inv"], c["img"], c["prev"], c["nxt"], c["mask"], c["K"], c["poses"], ssim_w=0.0, g_photo=1.0,
                                      g_smooth=0.0, automask=automask, reduce_op=reduce_op, padding_mode=padding_mode)
    if f"{tag}.raises" in z.files:
        with pytest.raises(IndexError):
            call()
        return
    r = call()
    key = lambda k: z[f"{tag}.{k}"]
    assert float(r["loss_photometric"]) == pytest.approx(float(key("loss_photometric")), rel=2e-5)
    assert float(r["loss_smoothness"]) == pytest.approx(float(key("loss_smoothness")), rel=2e-5)
    for i in range(3):
        _grad_close(r["d_inv"][i], key(f"dphot_dinv{i}"), name=f"{tag}/dphot_dinv{i}")
    ref_p = key("dphot_dposes")
    np.testing.assert_allclose(r["d_poses"], ref_p, rtol=5e-3, atol=5e-3 * np.abs(ref_p).max() + 1e-9)
