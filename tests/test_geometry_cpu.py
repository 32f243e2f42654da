"""`mgnet` import name + the host-side part of `mgnet.geometry` against outputs of the reference's own package
(tests/golden/geometry.npz, made by tests/golden/make_golden_geometry.py)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

REF_CFG = "/root/reference/configs"


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "geometry.npz"))


def test_mgnet_import_name():
    """mgnet/__init__.py:1 `from mgnet import add_mgnet_config`; mgnet/geometry/__init__.py:1-16 names"""
    import mgnet
    import mgnet_amd
    from mgnet import add_mgnet_config
    from mgnet.geometry import (Camera, Pose, calc_smoothness, construct_K, euler2mat, gradient_x, gradient_y,  # noqa: F401
                                image_grid, interpolate_image, inv2depth, invert_pose, match_scales, meshgrid,
                                pose_vec2mat, same_shape, scale_intrinsics, view_synthesis)
    import mgnet.data, mgnet.evaluation, mgnet.modeling, mgnet.modeling.loss, mgnet.postprocessing, mgnet.solver  # noqa: E401,F401
    assert add_mgnet_config is mgnet_amd.add_mgnet_config
    assert mgnet.modeling is mgnet_amd.modeling and mgnet.modeling.loss is mgnet_amd.modeling.loss
    assert mgnet.geometry is mgnet_amd.geometry
    assert mgnet_amd.modeling.__spec__.name == "mgnet_amd.modeling"      # the alias leaves the real module untouched
    from mgnet.modeling import MGNet, MGNetSemSegHead, MultiViewPhotometricLoss  # noqa: F401
    with pytest.raises(ModuleNotFoundError):
        import mgnet.does_not_exist  # noqa: F401


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference checkout not present (GPU box)")
def test_reference_yamls_through_mgnet_name():
    from mgnet import add_mgnet_config, get_cfg
    files = sorted(glob.glob(os.path.join(REF_CFG, "MGNet-*.yaml")))
    assert len(files) == 5
    for f in files:
        cfg = get_cfg()
        add_mgnet_config(cfg)
        cfg.merge_from_file(f)
        assert cfg.MODEL.META_ARCHITECTURE == "MGNet"


def test_small_matrix_functions(gold):
    from mgnet.geometry import Camera, Pose, construct_K, euler2mat, invert_pose, pose_vec2mat, scale_intrinsics
    t = torch.from_numpy
    vec, vec_cam, K = t(gold["in_vec"]), t(gold["in_vec_cam"]), t(gold["in_K"])
    tol = dict(rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(euler2mat(vec[:, 3:]).numpy(), gold["out_euler2mat"], **tol)
    np.testing.assert_allclose(pose_vec2mat(vec).numpy(), gold["out_pose_vec2mat"], **tol)
    assert pose_vec2mat(vec, None) is vec
    with pytest.raises(ValueError):
        pose_vec2mat(vec, "quaternion")
    P = Pose.from_vec(vec, "euler")
    np.testing.assert_allclose(P.item().numpy(), gold["out_pose_mat"], **tol)
    np.testing.assert_allclose(invert_pose(P.item()).numpy(), gold["out_invert_pose"], **tol)
    np.testing.assert_allclose(P.inverse().item().numpy(), gold["out_invert_pose"], **tol)
    np.testing.assert_allclose((P @ Pose.from_vec(vec_cam, "euler")).item().numpy(), gold["out_pose_compose"], **tol)
    assert len(P) == 2 and tuple(P.shape) == (2, 4, 4) and len(Pose.identity(3)) == 3
    with pytest.raises(NotImplementedError):
        P @ 3
    with pytest.raises(ValueError):
        P @ torch.zeros(2, 4)
    np.testing.assert_array_equal(scale_intrinsics(K.clone(), 0.5, 0.25).numpy(), gold["out_scale_intrinsics"])
    np.testing.assert_array_equal(construct_K(100.0, 110.0, 17.5, 9.25).numpy(), gold["out_construct_K"])
    cam = Camera(K.clone())
    np.testing.assert_array_equal(cam.Kinv.numpy(), gold["out_Kinv"])          # incl. the un-inverted skew entry
    np.testing.assert_array_equal(cam.scaled(0.5).K.numpy(), gold["out_scaled_K"])
    assert cam.scaled(1.0) is cam and len(cam) == 2
    np.testing.assert_array_equal(cam.Twc.item().numpy(), np.tile(np.eye(4, dtype=np.float32), (2, 1, 1)))
    # points through a Pose (pose.py:77-83)
    pts = torch.randn(2, 3, 4, 5)
    ref = (P.item()[:, :3, :3] @ pts.view(2, 3, -1) + P.item()[:, :3, 3:]).view(2, 3, 4, 5)
    np.testing.assert_allclose((P @ pts).numpy(), ref.numpy(), rtol=1e-6, atol=1e-6)


def test_tensor_helpers(gold):
    from mgnet.geometry import (calc_smoothness, gradient_x, gradient_y, image_grid, interpolate_image, inv2depth,
                                match_scales, same_shape)
    t = torch.from_numpy
    inv = [t(gold["in_inv0"]), t(gold["in_inv1"])]
    ref = t(gold["in_ref"])
    np.testing.assert_array_equal(inv2depth(inv[0]).numpy(), gold["out_inv2depth0"])
    assert isinstance(inv2depth(tuple(inv)), list) and len(inv2depth(inv)) == 2
    sx, sy = calc_smoothness(inv, ref, 2)
    for i in range(2):
        np.testing.assert_allclose(sx[i].numpy(), gold[f"out_smooth_x{i}"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(sy[i].numpy(), gold[f"out_smooth_y{i}"], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(gradient_x(ref).numpy(), gold["out_gradient_x"])
    np.testing.assert_array_equal(gradient_y(ref).numpy(), gold["out_gradient_y"])
    np.testing.assert_array_equal(image_grid(1, 5, 7, torch.float32, torch.device("cpu"), normalized=False).numpy(), gold["out_image_grid"])
    np.testing.assert_allclose(image_grid(1, 5, 7, torch.float32, torch.device("cpu"), normalized=True).numpy(),
                               gold["out_image_grid_norm"], rtol=0, atol=1e-7)
    B, _, H, W = ref.shape
    ms = match_scales(ref, [torch.zeros(B, 1, H // 2, W // 2), torch.zeros(B, 1, H, W)], 2)
    np.testing.assert_allclose(ms[0].numpy(), gold["out_match_scales0"], rtol=1e-6, atol=1e-7)
    assert ms[1] is ref and bool(gold["out_match_scales1_is_same"])
    assert interpolate_image(ref, (B, 3, H, W)) is ref
    assert same_shape((1, 2), (1, 2)) and not same_shape((1, 2), (1, 2, 3)) and not same_shape((1, 2), (1, 3))


def test_per_pixel_stages_refuse_cpu_tensors():
    """no CPU fallback behind the HIP-backed functions"""
    from mgnet.geometry import Camera, view_synthesis
    K = torch.eye(3).repeat(1, 1, 1)
    cam = Camera(K)
    with pytest.raises(RuntimeError):
        cam.reconstruct(torch.ones(1, 1, 4, 4))
    with pytest.raises(RuntimeError):
        cam.project(torch.ones(1, 3, 4, 4))
    with pytest.raises(RuntimeError):
        view_synthesis(torch.ones(1, 3, 4, 4), torch.ones(1, 1, 4, 4), cam, cam)
