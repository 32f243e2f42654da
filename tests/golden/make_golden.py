#!/usr/bin/env python3
"""Generate golden vectors for the reprojection-loss hot path FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference).  The reference's Python never
travels to the GPU box; only the .npz fixtures written next to this script do.

Recipe = SURVEY.md Appendix E: stub the parent packages so `mgnet/__init__.py` and
`mgnet/modeling/__init__.py` (which need detectron2) are not executed, import
`mgnet.geometry` + `mgnet.modeling.loss` (torch-only), apply two harness-side shims for the
CUDA-isms at loss.py:160-161 and loss.py:60.  Nothing in the reference is edited or copied.

Usage:  python tests/golden/make_golden.py            (rewrites tests/golden/*.npz)
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/mgnet"
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    for name, path in (("mgnet", REF), ("mgnet.modeling", REF + "/modeling")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    import mgnet.geometry as G  # noqa
    import mgnet.modeling.loss as L  # noqa

    _to = G.Pose.to
    G.Pose.to = lambda s, *a, **k: s if (a and isinstance(a[0], int) and a[0] < 0) else _to(s, *a, **k)
    torch.Tensor.cuda = lambda s, *a, **k: s
    return G, L


# ------------------------------------------------------------------------------------------
# deterministic input builders (numpy RandomState: stable across library versions)
# ------------------------------------------------------------------------------------------
def make_K(B, H, W, fx_rel=0.58, fy_rel=1.92, jitter=None):
    K = np.zeros((B, 4, 4), np.float32)
    for b in range(B):
        s = 1.0 + (0.03 * b if jitter else 0.0)
        K[b] = np.array(
            [[fx_rel * W * s, 0, 0.5 * W - 0.5 * b, 0], [0, fy_rel * H * s, 0.5 * H + 0.25 * b, 0], [0, 0, 1, 0], [0, 0, 0, 1]],
            np.float32,
        )
    return K


def smooth_field(rs, B, C, H, W, n=6):
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    out = np.zeros((B, C, H, W), np.float64)
    for b in range(B):
        for c in range(C):
            acc = np.zeros((H, W))
            for _ in range(n):
                fx, fy = rs.uniform(0.02, 0.35, 2)
                ph = rs.uniform(0, 2 * np.pi)
                acc += rs.uniform(0.3, 1.0) * np.sin(fx * xx + fy * yy + ph)
            acc = (acc - acc.min()) / (acc.max() - acc.min() + 1e-12)
            out[b, c] = acc
    return out.astype(np.float32)


def build_case(name):
    rs = np.random.RandomState(sum(map(ord, name)) * 7919 % (2**31))
    c = {}
    if name == "rand_small":
        B, H, W = 2, 24, 40
        c["inv"] = [rs.uniform(0.05, 1.95, (B, 1, H, W)).astype(np.float32) for _ in range(3)]
        c["img"], c["prev"], c["nxt"] = [rs.uniform(0, 1, (B, 3, H, W)).astype(np.float32) for _ in range(3)]
        c["poses"] = (0.02 * rs.randn(B, 2, 6)).astype(np.float32)
        c["mask"] = rs.uniform(0, 1, (B, 1, H, W)) > 0.1
        c["K"] = make_K(B, H, W, jitter=True)
    elif name == "smooth":
        B, H, W = 2, 32, 64
        base = smooth_field(rs, B, 3, H + 8, W + 8)
        c["img"] = base[:, :, 4:-4, 4:-4].copy()
        c["prev"] = base[:, :, 3:-5, 1:-7].copy()
        c["nxt"] = base[:, :, 5:-3, 7:-1].copy()
        c["inv"] = [(0.1 + 1.8 * smooth_field(rs, B, 1, H, W, n=3)) for _ in range(3)]
        c["poses"] = (0.01 * rs.randn(B, 2, 6)).astype(np.float32)
        c["poses"][:, :, 0] += np.array([0.3, -0.3], np.float32)
        c["mask"] = rs.uniform(0, 1, (B, 1, H, W)) > 0.1
        c["K"] = make_K(B, H, W)
    elif name == "identity_pose":
        B, H, W = 1, 16, 32
        c["inv"] = [rs.uniform(0.05, 1.95, (B, 1, H, W)).astype(np.float32) for _ in range(3)]
        c["img"] = smooth_field(rs, B, 3, H, W)
        c["prev"] = smooth_field(rs, B, 3, H, W)
        c["nxt"] = c["img"].copy()  # unwarped loss of `next` is exactly 0 -> exercises min ties / automask
        c["poses"] = np.zeros((B, 2, 6), np.float32)
        c["mask"] = np.ones((B, 1, H, W), bool)
        c["K"] = make_K(B, H, W)
    elif name == "oob_clamp":
        # large motion: many samples out of bounds, points behind the camera (z<1e-5 clamp),
        # inverse depths below the 1e-6 clamp and exactly 0
        B, H, W = 2, 20, 36
        inv = [rs.uniform(0.05, 1.95, (B, 1, H, W)).astype(np.float32) for _ in range(3)]
        inv[0][:, :, :4, :6] = 0.0
        inv[1][:, :, 5:8, :] = 5e-7
        inv[2][:, :, :, 30:] = -0.3
        c["inv"] = inv
        c["img"], c["prev"], c["nxt"] = [rs.uniform(0, 1, (B, 3, H, W)).astype(np.float32) for _ in range(3)]
        poses = (0.05 * rs.randn(B, 2, 6)).astype(np.float32)
        poses[0, 0, :3] = [2.5, -0.7, -4.0]  # tz=-4: depth<4 ends up behind the camera
        poses[1, 1, :3] = [-6.0, 1.0, 0.5]
        poses[1, 0, 3:] = [0.2, -0.35, 0.4]
        c["poses"] = poses
        c["mask"] = rs.uniform(0, 1, (B, 1, H, W)) > 0.3
        c["K"] = make_K(B, H, W, jitter=True)
    elif name == "no_mask_odd":
        # mask key absent (reference builds all-ones), odd sizes, skewed intrinsics
        B, H, W = 3, 17, 29
        c["inv"] = [rs.uniform(0.05, 1.95, (B, 1, H, W)).astype(np.float32) for _ in range(3)]
        c["img"], c["prev"], c["nxt"] = [smooth_field(rs, B, 3, H, W) for _ in range(3)]
        c["poses"] = (0.03 * rs.randn(B, 2, 6)).astype(np.float32)
        c["mask"] = None
        K = make_K(B, H, W, jitter=True)
        K[:, 0, 1] = 0.37  # skew: Camera.Kinv (camera.py:74-81) copies K[0,1] verbatim -- keep that quirk
        c["K"] = K
    elif name == "survey_192x640":
        B, H, W = 2, 192, 640
        c["inv"] = [rs.uniform(0.05, 1.95, (B, 1, H, W)).astype(np.float32) for _ in range(3)]
        c["img"], c["prev"], c["nxt"] = [rs.uniform(0, 1, (B, 3, H, W)).astype(np.float32) for _ in range(3)]
        c["poses"] = (0.01 * rs.randn(B, 2, 6)).astype(np.float32)
        c["mask"] = rs.uniform(0, 1, (B, 1, H, W)) > 0.1
        c["K"] = make_K(B, H, W)
    else:
        raise KeyError(name)
    return c


CASES = ["rand_small", "smooth", "identity_pose", "oob_clamp", "no_mask_odd", "survey_192x640"]
# cases whose inputs are regenerated from the seed at test time instead of being stored (size)
SEED_ONLY = {"survey_192x640"}


def run_reference(G, L, c, stages=True):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    inv = [t(a).requires_grad_(True) for a in c["inv"]]
    poses = t(c["poses"]).requires_grad_(True)
    img, prev, nxt, K = t(c["img"]), t(c["prev"]), t(c["nxt"]), t(c["K"])
    targets = {"image_orig": img, "image_prev_orig": prev, "image_next_orig": nxt, "camera_matrix": K}
    if c["mask"] is not None:
        targets["reprojection_mask"] = t(c["mask"])
    loss = L.MultiViewPhotometricLoss(0.85, 1.0, 0.001, True, "min", "zeros")  # config.py:109-117 defaults
    out = loss({"depth": inv, "poses": poses}, targets)
    # independent upstream weights so that both scalars' gradients are pinned separately
    g_p, g_s = 1.0, 1.0
    res = {
        "loss_photometric": out["loss_photometric"].detach().numpy(),
        "loss_smoothness": out["loss_smoothness"].detach().numpy(),
    }
    gp = torch.autograd.grad(out["loss_photometric"], inv + [poses], retain_graph=True, allow_unused=True)
    gs = torch.autograd.grad(out["loss_smoothness"], inv + [poses], allow_unused=True)
    for i in range(3):
        res[f"dphot_dinv{i}"] = gp[i].numpy()
        res[f"dsmooth_dinv{i}"] = gs[i].numpy()
    res["dphot_dposes"] = gp[3].numpy()
    assert gs[3] is None  # smoothness does not depend on the poses
    if stages:
        with torch.no_grad():
            Kc = K[:, :3, :3].float()
            depths = [G.inv2depth(x.detach()) for x in inv]
            cam = G.Camera(K=Kc)
            mask = t(c["mask"]) if c["mask"] is not None else torch.ones_like(inv[0], dtype=torch.bool)
            plist = [[] for _ in range(3)]
            for j, ref in enumerate((prev, nxt)):
                pose = G.Pose.from_vec(poses.detach()[:, j].float(), "euler")
                res[f"pose_mat{j}"] = pose.mat.numpy().copy()
                ref_cam = G.Camera(K=Kc, Tcw=pose)
                warped = [G.view_synthesis(ref, depths[i], ref_cam, cam, padding_mode="zeros") for i in range(3)]
                ph = loss.calc_photometric_loss(warped, [img] * 3)
                un = loss.calc_photometric_loss([ref], [img])[0]
                res[f"unwarped{j}"] = un.numpy()
                for i in range(3):
                    res[f"warped{j}_{i}"] = warped[i].numpy()
                    res[f"photo{j}_{i}"] = ph[i].numpy()
                    plist[i] += [ph[i], un]
            for i in range(3):
                res[f"minmap{i}"] = torch.cat(plist[i], 1).min(1, True)[0].numpy()
            sx, sy = G.calc_smoothness([x.detach() for x in inv], img, 3)
            for i in range(3):
                res[f"smooth_x{i}"] = sx[i].numpy()
                res[f"smooth_y{i}"] = sy[i].numpy()
    return res


def kats(G, L):
    """Known-answer facts of the reference, captured as numbers."""
    k = {}
    ang = torch.tensor([[0.3, -0.2, 0.5], [0.0, 0.0, 0.0], [-1.1, 0.7, 2.0]])
    k["euler_in"] = ang.numpy()
    k["euler_out"] = G.euler2mat(ang).numpy()
    vec = torch.tensor([[0.1, -0.2, 0.3, 0.3, -0.2, 0.5]])
    k["vec_in"] = vec.numpy()
    k["vec_mat"] = G.Pose.from_vec(vec, "euler").mat.numpy()
    k["vec_mat_inv"] = G.Pose.from_vec(vec, "euler").inverse().mat.numpy()
    x = torch.rand(1, 3, 6, 7, generator=torch.Generator().manual_seed(3))
    k["ssim_xx"] = L.MultiViewPhotometricLoss.ssim(x, x).numpy()
    k["ssim_01"] = L.MultiViewPhotometricLoss.ssim(torch.zeros(1, 1, 5, 5), torch.ones(1, 1, 5, 5)).numpy()
    k["inv2depth_in"] = np.array([0.0, 1e-7, 1e-6, 0.5, 2.0], np.float32)
    k["inv2depth_out"] = G.inv2depth(torch.from_numpy(k["inv2depth_in"])).numpy()
    K = torch.tensor([[[100.0, 0, 50.0], [0, 120.0, 30.0], [0, 0, 1]]])
    k["scale_K_in"] = K.numpy()
    k["scale_K_out"] = G.scale_intrinsics(K.clone(), 0.5, 0.25).numpy()
    k["Kinv"] = G.Camera(K=K).Kinv.numpy()
    return k


def ce_cases(L):
    """OhemCE / DeepLabCE (loss.py:9-81) on small logits -> value + dlogits (row N9)."""
    res = {}
    rs = np.random.RandomState(77)
    B, C, H, W = 2, 7, 12, 20
    logits = (2.0 * rs.randn(B, C, H, W)).astype(np.float32)
    labels = rs.randint(0, C, (B, H, W)).astype(np.int64)
    labels[rs.uniform(size=labels.shape) < 0.1] = 255
    weights = np.where(rs.uniform(size=(B, H, W)) < 0.2, 3.0, 1.0).astype(np.float32)
    res["logits"], res["labels"], res["weights"] = logits, labels, weights
    for tag, n_min, thr in (("ohem_top", 100, 0.7), ("ohem_thr", 20, 0.2), ("ohem_top_hi", 300, 0.95)):
        lg = torch.from_numpy(logits).requires_grad_(True)
        crit = L.OhemCE(ignore_label=255, ohem_threshold=thr, n_min=n_min)
        v = crit(lg, torch.from_numpy(labels), torch.from_numpy(weights))
        (g,) = torch.autograd.grad(v, lg)
        res[f"{tag}_cfg"] = np.array([n_min, thr], np.float64)
        res[f"{tag}_val"], res[f"{tag}_grad"] = v.detach().numpy(), g.numpy()
    for tag, k in (("dl_all", 1.0), ("dl_top", 0.2)):
        lg = torch.from_numpy(logits).requires_grad_(True)
        crit = L.DeepLabCE(ignore_label=255, top_k_percent_pixels=k)
        v = crit(lg, torch.from_numpy(labels), torch.from_numpy(weights))
        (g,) = torch.autograd.grad(v, lg)
        res[f"{tag}_cfg"] = np.array([k], np.float64)
        res[f"{tag}_val"], res[f"{tag}_grad"] = v.detach().numpy(), g.numpy()
    return res


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)  # deterministic reduction order for the fixtures
    G, L = import_reference()
    for name in CASES:
        c = build_case(name)
        seed_only = name in SEED_ONLY
        res = run_reference(G, L, c, stages=not seed_only)
        blob = {}
        if seed_only:
            # keep only scalars, pose grads and a strided sample of the dense grads
            keep = {k: v for k, v in res.items() if k.startswith("loss_") or k == "dphot_dposes"}
            for i in range(3):
                keep[f"dphot_dinv{i}_s"] = res[f"dphot_dinv{i}"][:, :, ::16, ::16].copy()
                keep[f"dsmooth_dinv{i}_s"] = res[f"dsmooth_dinv{i}"][:, :, ::16, ::16].copy()
                keep[f"dphot_dinv{i}_abssum"] = np.abs(res[f"dphot_dinv{i}"].astype(np.float64)).sum()
                keep[f"dsmooth_dinv{i}_abssum"] = np.abs(res[f"dsmooth_dinv{i}"].astype(np.float64)).sum()
            blob.update({"out_" + k: v for k, v in keep.items()})
        else:
            for i in range(3):
                blob[f"in_inv{i}"] = c["inv"][i]
            for k in ("img", "prev", "nxt", "poses", "K"):
                blob["in_" + k] = c[k]
            if c["mask"] is not None:
                blob["in_mask"] = c["mask"]
            blob.update({"out_" + k: v for k, v in res.items()})
        path = os.path.join(OUT, f"reproj_{name}.npz")
        np.savez_compressed(path, **blob)
        print(f"{name:16s} Lp={float(res['loss_photometric']):.7f} Ls={float(res['loss_smoothness']):.7e} "
              f"-> {os.path.getsize(path)/1024:.0f} KiB")
    np.savez_compressed(os.path.join(OUT, "kats.npz"), **kats(G, L))
    np.savez_compressed(os.path.join(OUT, "ce_losses.npz"), **ce_cases(L))
    print("kats + ce_losses written")


if __name__ == "__main__":
    main()
