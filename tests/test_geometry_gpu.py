"""The HIP-backed per-pixel stages of `mgnet.geometry` (csrc/geometry.hip through the C-ABI) against outputs and autograd
gradients of the reference's own package (tests/golden/geometry.npz) and against the per-stage tensors of the loss fixtures
(tests/golden/reproj_*.npz: warped{j}_{i}, pose_mat{j})."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPROJ_CASES, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "geometry.npz"))


def _t(a, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).requires_grad_(grad)


def _close(got, ref, rtol, atol_rel, what, frac=0.0):
    """|got-ref| <= rtol*|ref| + atol_rel*max|ref| on all but `frac` of the elements"""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    bad = np.abs(got - ref) > rtol * np.abs(ref) + atol_rel * np.abs(ref).max()
    assert bad.mean() <= frac, (what, float(bad.mean()), float(np.abs(got - ref).max()), float(np.abs(ref).max()))


@pytest.mark.parametrize("frame", ["c", "w"])
def test_reconstruct_and_project(gold, frame):
    from mgnet.geometry import Camera, Pose
    K = _t(gold["in_K"])
    d = _t(gold["in_depth"], True)
    v = _t(gold["in_vec_cam"], True)
    cam = Camera(K.clone(), Tcw=Pose.from_vec(v, "euler"))
    pts = cam.reconstruct(d, frame=frame)
    (pts * _t(gold["in_g_pts"])).sum().backward()
    _close(pts.detach().cpu(), gold[f"out_reconstruct_{frame}"], 1e-5, 1e-6, "points")
    _close(d.grad.cpu(), gold[f"out_reconstruct_{frame}_ddepth"], 1e-4, 1e-6, "d depth")
    if frame == "w":
        _close(v.grad.cpu(), gold[f"out_reconstruct_{frame}_dvec"], 1e-3, 1e-4, "d pose vec")
    X = _t(gold[f"out_reconstruct_{frame}"], True)
    v2 = _t(gold["in_vec"], True)
    cam2 = Camera(K.clone(), Tcw=Pose.from_vec(v2, "euler"))
    co = cam2.project(X, frame=frame)
    (co * _t(gold["in_g_co"])).sum().backward()
    _close(co.detach().cpu(), gold[f"out_project_{frame}"], 1e-5, 1e-6, "coords")
    _close(X.grad.cpu(), gold[f"out_project_{frame}_dX"], 1e-4, 1e-6, "d points")
    if frame == "w":
        _close(v2.grad.cpu(), gold[f"out_project_{frame}_dvec"], 1e-3, 1e-4, "d pose vec")
    with pytest.raises(ValueError):
        cam.reconstruct(d, frame="x")
    with pytest.raises(ValueError):
        cam.project(X, frame="x")


def test_view_synthesis_matches_reference_with_gradients(gold):
    from mgnet.geometry import Camera, Pose, view_synthesis
    K = _t(gold["in_K"])
    d = _t(gold["in_depth"], True)
    v, vc = _t(gold["in_vec"], True), _t(gold["in_vec_cam"], True)
    cam = Camera(K.clone(), Tcw=Pose.from_vec(vc, "euler"))
    ref_cam = Camera(K.clone(), Tcw=Pose.from_vec(v, "euler"))
    warped = view_synthesis(_t(gold["in_ref"]), d, ref_cam, cam)
    (warped * _t(gold["in_g_img"])).sum().backward()
    # coordinate round-off moves a sample by ~1e-5 px (abs 1e-5 * max(1, W/64)); the images here are white noise
    # (slope up to 1 per pixel), and a pixel whose sample position sits on an integer switches its corner set
    _close(warped.detach().cpu(), gold["out_view_synthesis"], 0, 2e-5, "warped", frac=0.002)
    _close(d.grad.cpu(), gold["out_view_synthesis_ddepth"], 1e-3, 1e-4, "d depth", frac=0.005)
    _close(v.grad.cpu(), gold["out_view_synthesis_dvec"], 2e-3, 2e-3, "d pose vec (ref cam)")
    _close(vc.grad.cpu(), gold["out_view_synthesis_dvec_cam"], 2e-3, 2e-3, "d pose vec (cam)")
    with pytest.raises(NotImplementedError):
        view_synthesis(_t(gold["in_ref"]), d, ref_cam, cam, padding_mode="border")
    with pytest.raises(NotImplementedError):
        view_synthesis(_t(gold["in_ref"]), d, ref_cam, cam, mode="nearest")
    with pytest.raises(NotImplementedError):          # the sampled image is data: no gradient defined for it
        view_synthesis(_t(gold["in_ref"], True), d, ref_cam, cam).sum().backward()


@pytest.mark.parametrize("case", REPROJ_CASES)
def test_view_synthesis_reproduces_loss_fixture_stages(case):
    """loss.py:156-167 warp_ref_image, stage by stage: Pose.from_vec -> Camera(K, Tcw) -> view_synthesis per scale"""
    from mgnet.geometry import Camera, Pose, inv2depth, view_synthesis
    ins, outs = load_golden("reproj_" + case)
    K = _t(ins["K"])[:, :3, :3].contiguous()
    W = ins["img"].shape[-1]
    for j, name in enumerate(("prev", "nxt")):
        pose = Pose.from_vec(_t(ins["poses"][:, j]), "euler")
        np.testing.assert_allclose(pose.item().cpu().numpy(), outs[f"pose_mat{j}"], rtol=1e-6, atol=1e-7)
        for i in range(3):
            if f"warped{j}_{i}" not in outs:
                continue
            depth = inv2depth(_t(ins[f"inv{i}"]))
            warped = view_synthesis(_t(ins[name]), depth, Camera(K, Tcw=pose), Camera(K))
            _close(warped.cpu(), outs[f"warped{j}_{i}"], 0, 2e-5 * max(1.0, W / 64), f"warped{j}_{i}", frac=0.003)


def test_view_synthesis_full_size_properties():
    """1024x2048 (C4 frame): identity pose + unit intrinsics map returns the image itself; a pure 1-pixel image-plane shift
    returns the shifted image with zeros where the source is outside (padding_mode='zeros')"""
    from mgnet.geometry import Camera, Pose, view_synthesis
    B, H, W = 2, 1024, 2048
    g = torch.Generator(device=DEV).manual_seed(0)
    img = torch.rand(B, 3, H, W, device=DEV, generator=g)
    depth = torch.rand(B, 1, H, W, device=DEV, generator=g) * 50 + 1
    K = torch.tensor([[2262.52, 0, 1096.98], [0, 2265.30, 513.137], [0, 0, 1]], device=DEV).repeat(B, 1, 1)
    same = view_synthesis(img, depth, Camera(K), Camera(K))
    assert float((same - img).abs().max()) < 5e-4           # |ix - u| <= ~2e-4 px round-off (u up to 2047) times slope <= 1
    depth1 = torch.ones(B, 1, H, W, device=DEV)
    vec = torch.zeros(B, 6, device=DEV)
    vec[:, 0] = 1.0 / 2262.52                               # X -> X + fx * tx / depth = one pixel at depth 1
    shifted = view_synthesis(img, depth1, Camera(K, Tcw=Pose.from_vec(vec, "euler")), Camera(K))
    assert float((shifted[..., :-1] - img[..., 1:]).abs().max()) < 2e-3
    assert float(shifted[..., -1].abs().max()) < 2e-3       # samples at x = W: both corners outside or weight ~ 0
