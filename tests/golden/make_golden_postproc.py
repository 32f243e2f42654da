"""Generates tests/golden/postproc_*.npz from the reference's OWN post-processing (mgnet/postprocessing/panoptic_post_proc.py
and depth_post_proc.py, imported unmodified).  Runs only where /root/reference exists (the build container):

    python tests/golden/make_golden_postproc.py

Harness-side shims for the CUDA-isms of those files (nothing in the reference is edited): `torch.linspace(..., device="cuda")`
(panoptic_post_proc.py:98-107) loses its device argument, `Tensor.cuda()` (:134, depth_post_proc.py:171) is the identity.
The parent packages are stubbed so that `mgnet/__init__.py` (detectron2) is not executed.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/mgnet"

PAN_KW = dict(num_thing_classes=8, last_stuff_id=10, label_divisor=1000, void_label=-1)


def import_reference():
    for name, path in (("mgnet", REF), ("mgnet.postprocessing", REF + "/postprocessing")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    _lin = torch.linspace

    def linspace(*a, **k):
        k.pop("device", None)
        return _lin(*a, **k)
    torch.linspace = linspace
    torch.Tensor.cuda = lambda s, *a, **k: s
    import mgnet.postprocessing.depth_post_proc as D  # noqa
    import mgnet.postprocessing.panoptic_post_proc as P  # noqa
    return P, D


# ---- deterministic inputs ---------------------------------------------------------------------------------------------
def pan_case(seed, H, W, n_inst=7, noise=1.5, integer_offsets=False, plateau=False, peak=0.9, all_stuff=False, rmax=None):
    """Semantic blocks + instances: heat map = max of Gaussians at the instance centres (+ optional flat plateaus, whose
    tied maxima all survive the NMS), offsets = centre - pixel + noise inside a disc around each centre, random elsewhere."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    blk = rs.randint(0, 11 if all_stuff else 19, size=((H + 7) // 8, (W + 7) // 8))
    sem = np.kron(blk, np.ones((8, 8), dtype=np.int64))[:H, :W].astype(np.int64)
    center = np.zeros((H, W), np.float32)
    off = (rs.randn(2, H, W) * 6).astype(np.float32)
    for _ in range(n_inst):
        cy, cx, r = rs.uniform(0, H), rs.uniform(0, W), rs.uniform(4, rmax or min(H, W) / 3)
        center = np.maximum(center, (peak * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 3.0 ** 2))).astype(np.float32))
        m = (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
        if not all_stuff:
            sem[m] = rs.randint(11, 19)
        off[0][m] = (cy - yy + noise * rs.randn(H, W))[m]
        off[1][m] = (cx - xx + noise * rs.randn(H, W))[m]
    if plateau:
        center[H // 3:H // 3 + 3, W // 4:W // 4 + 5] = 0.75      # flat top: every pixel equals its window maximum
        center[H // 2, :] = np.maximum(center[H // 2, :], 0.5)   # a ridge
    if integer_offsets:
        off = np.round(off)                                      # exact distance ties between centres
    center += (rs.rand(H, W) * 0.02).astype(np.float32)
    return sem, center.astype(np.float32), off.astype(np.float32)


PAN_CASES = {
    "clean": (dict(seed=1, H=48, W=80), dict(stuff_area=64, threshold=0.3, nms_kernel=7)),
    "ties": (dict(seed=2, H=40, W=64, integer_offsets=True, plateau=True), dict(stuff_area=32, threshold=0.3, nms_kernel=7)),
    "no_centers": (dict(seed=3, H=24, W=40, peak=0.2), dict(stuff_area=16, threshold=0.3, nms_kernel=7)),
    "no_things": (dict(seed=4, H=24, W=40, all_stuff=True), dict(stuff_area=70, threshold=0.3, nms_kernel=7)),
    "ragged_k3": (dict(seed=5, H=37, W=53, n_inst=12, noise=4.0), dict(stuff_area=100, threshold=0.1, nms_kernel=3)),
}


def depth_case(seed, H, W):
    """A ground plane seen by a pin-hole camera 1.3 (unscaled) above it + a frontal wall + noise; panoptic labels: road
    (0) on most of the plane, sky (10000), a car instance."""
    rs = np.random.RandomState(seed)
    K = np.array([[0.9 * W, 0, W / 2 - 0.5], [0, 1.1 * W, H * 0.45], [0, 0, 1]], np.float32)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    ray_y = (yy - K[1, 2]) / K[1, 1]
    cam_h = 1.3
    depth = np.where(ray_y > 0.02, cam_h / np.maximum(ray_y, 0.02), 40.0).astype(np.float32)
    depth = np.minimum(depth, 40.0) * (1 + 0.01 * rs.randn(H, W)).astype(np.float32)
    depth[: H // 3, W // 3: W // 2] = 12.0
    pan = np.full((H, W), 2000, np.int64)
    pan[ray_y > 0.05] = 0
    pan[: H // 5] = 10000
    pan[H // 2: H // 2 + 6, W // 2: W // 2 + 9] = 13001
    return depth.astype(np.float32), K, pan


DEPTH_CASES = {
    "dgc_panoptic": dict(seed=1, H=40, W=72, use_dgc=True, with_pan=True),
    "dgc_normals": dict(seed=2, H=36, W=64, use_dgc=True, with_pan=False),
    "no_dgc": dict(seed=3, H=24, W=40, use_dgc=False, with_pan=True),
}
DEPTH_KW = dict(road_class_id=0, depth_filter_class_ids=[10000, 2000])


def main():
    P, D = import_reference()
    for name, (ikw, pkw) in PAN_CASES.items():
        sem, center, off = pan_case(**ikw)
        out = P.get_panoptic_prediction(torch.from_numpy(sem)[None].clone(), torch.from_numpy(center)[None].clone(),
                                        torch.from_numpy(off).clone(), **PAN_KW, **pkw)
        assert out.dtype == torch.int64 and tuple(out.shape) == sem.shape
        path = os.path.join(HERE, f"postproc_pan_{name}.npz")
        np.savez_compressed(path, sem=sem.astype(np.uint8), center=center, offsets=off, panoptic=out.numpy().astype(np.int32),
                            kw_keys=np.array(sorted(pkw)), kw_vals=np.array([float(pkw[k]) for k in sorted(pkw)]))
        ids = np.unique(out.numpy())
        print(name, sem.shape, "ids:", len(ids), "instances:", int((ids >= 11000).sum()), "void px:", int((out == -1).sum()),
              os.path.getsize(path), "bytes")
    for name, c in DEPTH_CASES.items():
        depth, K, pan = depth_case(c["seed"], c["H"], c["W"])
        d, xyz = D.get_depth_prediction(torch.from_numpy(depth)[None, None].clone(), c["use_dgc"],
                                        camera_matrix=torch.from_numpy(K)[None], real_camera_height=torch.tensor([1.65]),
                                        panoptic_seg=torch.from_numpy(pan) if c["with_pan"] else None,
                                        **(DEPTH_KW if c["with_pan"] else {}))
        path = os.path.join(HERE, f"postproc_depth_{name}.npz")
        np.savez_compressed(path, depth_in=depth, K=K, panoptic=pan.astype(np.int32), use_dgc=np.array(c["use_dgc"]),
                            with_pan=np.array(c["with_pan"]), depth=d.numpy(),
                            xyz=xyz.numpy() if xyz is not None else np.zeros((0,), np.float32))
        print(name, depth.shape, "scale:", float(d.flatten()[-1] / depth.flatten()[-1]) if d.flatten()[-1] != 0 else None,
              os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("reference not present: fixtures can only be regenerated in the build container")
    main()
