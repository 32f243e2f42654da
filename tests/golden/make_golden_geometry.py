#!/usr/bin/env python3
"""Golden vectors for the public functions of `mgnet.geometry`, produced by THE REFERENCE's own package (imported as in
make_golden.py; runs only in the build container).  Output: tests/golden/geometry.npz -- inputs, every function's result,
and torch-autograd gradients of the three per-pixel stages with respect to depth / points / the pose vector.

Usage:  python tests/golden/make_golden_geometry.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402


def main():
    G, _ = make_golden.import_reference()
    rs = np.random.RandomState(20261002)
    B, H, W = 2, 20, 36
    out = {}
    K = make_golden.make_K(B, H, W, jitter=True)[:, :3, :3].copy()
    K[1, 0, 1] = 0.3                                            # a skew entry: the reference's Kinv leaves it un-inverted
    vec = (0.05 * rs.randn(B, 6)).astype(np.float32)
    vec_cam = (0.03 * rs.randn(B, 6)).astype(np.float32)        # a non-identity pose for the target camera as well
    depth = rs.uniform(0.6, 20.0, (B, 1, H, W)).astype(np.float32)
    depth[0, 0, 3, 4] = 1e-7
    ref = rs.uniform(0, 1, (B, 3, H, W)).astype(np.float32)
    g_img = rs.randn(B, 3, H, W).astype(np.float32)
    g_pts = rs.randn(B, 3, H, W).astype(np.float32)
    g_co = rs.randn(B, H, W, 2).astype(np.float32)
    out.update(in_K=K, in_vec=vec, in_vec_cam=vec_cam, in_depth=depth, in_ref=ref, in_g_img=g_img, in_g_pts=g_pts, in_g_co=g_co)

    t = torch.from_numpy
    # --- small-matrix helpers ---
    out["out_euler2mat"] = G.euler2mat(t(vec[:, 3:])).numpy()
    out["out_pose_vec2mat"] = G.pose_vec2mat(t(vec)).numpy()
    P = G.Pose.from_vec(t(vec), "euler")
    out["out_pose_mat"] = P.item().numpy()
    out["out_invert_pose"] = G.invert_pose(P.item()).numpy()
    out["out_pose_compose"] = (P @ G.Pose.from_vec(t(vec_cam), "euler")).item().numpy()
    out["out_scale_intrinsics"] = G.scale_intrinsics(t(K.copy()), 0.5, 0.25).numpy()
    out["out_construct_K"] = G.construct_K(100.0, 110.0, 17.5, 9.25).numpy()
    cam0 = G.Camera(t(K.copy()))
    out["out_Kinv"] = cam0.Kinv.numpy()
    out["out_scaled_K"] = cam0.scaled(0.5).K.numpy()
    # --- tensor helpers ---
    inv = [t(rs.uniform(0.0, 2.0, (B, 1, H, W)).astype(np.float32)) for _ in range(2)]
    inv[0][0, 0, 0, :3] = torch.tensor([0.0, 5e-7, -1.0])
    out["in_inv0"], out["in_inv1"] = inv[0].numpy(), inv[1].numpy()
    out["out_inv2depth0"] = G.inv2depth(inv[0]).numpy()
    sx, sy = G.calc_smoothness(inv, t(ref), 2)
    for i in range(2):
        out[f"out_smooth_x{i}"], out[f"out_smooth_y{i}"] = sx[i].numpy(), sy[i].numpy()
    out["out_gradient_x"], out["out_gradient_y"] = G.gradient_x(t(ref)).numpy(), G.gradient_y(t(ref)).numpy()
    out["out_image_grid"] = G.image_grid(1, 5, 7, torch.float32, torch.device("cpu"), normalized=False).numpy()
    out["out_image_grid_norm"] = G.image_grid(1, 5, 7, torch.float32, torch.device("cpu"), normalized=True).numpy()
    ms = G.match_scales(t(ref), [torch.zeros(B, 1, H // 2, W // 2), torch.zeros(B, 1, H, W)], 2)
    out["out_match_scales0"], out["out_match_scales1_is_same"] = ms[0].numpy(), np.array(ms[1].shape == ref.shape)
    # --- per-pixel stages with gradients ---
    for frame in ("c", "w"):
        d = t(depth).requires_grad_(True)
        v = t(vec_cam).requires_grad_(True)
        cam = G.Camera(t(K.copy()), Tcw=G.Pose.from_vec(v, "euler"))
        pts = cam.reconstruct(d, frame=frame)
        (pts * t(g_pts)).sum().backward()
        out[f"out_reconstruct_{frame}"] = pts.detach().numpy()
        out[f"out_reconstruct_{frame}_ddepth"] = d.grad.numpy()
        out[f"out_reconstruct_{frame}_dvec"] = v.grad.numpy() if v.grad is not None else np.zeros_like(vec)
        X = pts.detach().clone().requires_grad_(True)
        v2 = t(vec).requires_grad_(True)
        cam2 = G.Camera(t(K.copy()), Tcw=G.Pose.from_vec(v2, "euler"))
        co = cam2.project(X, frame=frame)
        (co * t(g_co)).sum().backward()
        out[f"out_project_{frame}"] = co.detach().numpy()
        out[f"out_project_{frame}_dX"] = X.grad.numpy()
        out[f"out_project_{frame}_dvec"] = v2.grad.numpy() if v2.grad is not None else np.zeros_like(vec)
    d = t(depth).requires_grad_(True)
    v = t(vec).requires_grad_(True)
    vc = t(vec_cam).requires_grad_(True)
    cam = G.Camera(t(K.copy()), Tcw=G.Pose.from_vec(vc, "euler"))
    ref_cam = G.Camera(t(K.copy()), Tcw=G.Pose.from_vec(v, "euler"))
    warped = G.view_synthesis(t(ref), d, ref_cam, cam)
    (warped * t(g_img)).sum().backward()
    out["out_view_synthesis"] = warped.detach().numpy()
    out["out_view_synthesis_ddepth"] = d.grad.numpy()
    out["out_view_synthesis_dvec"] = v.grad.numpy()
    out["out_view_synthesis_dvec_cam"] = vc.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "geometry.npz"), **out)
    print("wrote geometry.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
