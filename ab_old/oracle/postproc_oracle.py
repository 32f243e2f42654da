"""oracle/postproc_oracle.py -- numpy restatement of MGNet's inference post-processing.

TEST INFRASTRUCTURE ONLY (never imported by mgnet_amd/).

Follows mgnet/postprocessing/panoptic_post_proc.py:9-147 (get_panoptic_prediction, _group_instances_and_fuse_logits) and
mgnet/postprocessing/depth_post_proc.py:11-185 (get_depth_prediction with the DGC scale recovery, surface normals, ground
mask) + mgnet/geometry/camera.py:107-141 (Camera.reconstruct, frame "c").

Parity status: **PINNED** by tests/golden/postproc_*.npz = outputs of those files imported unmodified in the build
container (tests/golden/make_golden_postproc.py; two harness-side shims for their hard-coded `cuda` device).

Written per PIXEL (label of the nearest centre, vote histogram, fuse) like the device kernels, not as the reference's
chain of masked tensor ops.
"""
import numpy as np


# ---- panoptic ---------------------------------------------------------------------------------------------------------
def center_points(center, threshold=0.3, nms_kernel=7):
    """panoptic_post_proc.py:53-59: F.threshold(x, thr, -1) -> max-pool NMS (stride 1, implicit -inf padding) -> survivors
    > 0, in row-major order (torch.nonzero)."""
    c = np.where(center > np.float32(threshold), center, np.float32(-1)).astype(np.float32)
    H, W = c.shape
    r = (nms_kernel - 1) // 2
    pad = np.full((H + 2 * r, W + 2 * r), -np.inf, dtype=np.float32)
    pad[r:r + H, r:r + W] = c
    mx = np.full((H, W), -np.inf, dtype=np.float32)
    for dy in range(nms_kernel):
        for dx in range(nms_kernel):
            np.maximum(mx, pad[dy:dy + H, dx:dx + W], out=mx)
    keep = (c == mx) & (c > 0)
    return np.argwhere(keep)   # [N, 2] (y, x), row-major


def panoptic_prediction(sem_seg, center, offsets, *, num_thing_classes, last_stuff_id, label_divisor, stuff_area, void_label,
                        threshold=0.3, nms_kernel=7):
    """sem_seg [H,W] integer labels, center [H,W] f32, offsets [2,H,W] f32 (dy, dx) -> panoptic [H,W] int64."""
    sem = np.asarray(sem_seg).astype(np.int64)
    H, W = sem.shape
    pts = center_points(np.asarray(center, dtype=np.float32), threshold, nms_kernel)
    thing = sem > last_stuff_id
    pan = sem.copy()
    if len(pts) and thing.any():
        yy, xx = np.nonzero(thing)                                         # row-major, like the masked ops of :109-127
        ly = np.asarray(offsets[0], dtype=np.float32)[thing] + yy.astype(np.float32)   # :108 offsets += xy (float32)
        lx = np.asarray(offsets[1], dtype=np.float32)[thing] + xx.astype(np.float32)
        best = np.zeros(len(yy), dtype=np.int64)
        bd = np.full(len(yy), np.inf, dtype=np.float32)
        for i, (py, px) in enumerate(pts):                                 # :125 argmin of the float32 L2 norm, first minimum
            dy, dx = np.float32(py) - ly, np.float32(px) - lx
            d = np.sqrt(dy * dy + dx * dx, dtype=np.float32)
            upd = d < bd
            best[upd], bd[upd] = i, d[upd]
        inst = best + 1
        n_inst = int(inst.max())                                           # :129 (trailing centres without pixels drop out)
        m0 = num_thing_classes + 1
        votes = np.zeros((n_inst, m0), dtype=np.int64)                     # :130-137 class voting
        np.add.at(votes, (inst - 1, sem[thing] - last_stuff_id), 1)
        cls = votes.argmax(1)                                              # first maximum
        pan[thing] = inst + (cls[inst - 1] + last_stuff_id) * label_divisor   # :138-144
    for k in range(last_stuff_id + 1):                                     # :64-66
        if (pan == k).sum() < stuff_area:
            pan[pan == k] = void_label
    m = (pan < label_divisor) & (pan != void_label)                        # :68-69
    pan[m] *= label_divisor
    return pan


# ---- depth --------------------------------------------------------------------------------------------------------------
def reconstruct(depth, K):
    """camera.py:107-141 frame 'c': Xc = (Kinv @ [u, v, 1]) * depth, Kinv in closed form (:74-81), float32."""
    H, W = depth.shape
    fx, fy, cx, cy = (np.float32(K[0, 0]), np.float32(K[1, 1]), np.float32(K[0, 2]), np.float32(K[1, 2]))
    one = np.float32(1)
    kinv = np.array([[one / fx, 0, -cx / fx], [0, one / fy, -cy / fy], [0, 0, 1]], dtype=np.float32)
    v, u = np.mgrid[0:H, 0:W].astype(np.float32)
    grid = np.stack([u, v, np.ones_like(u)], 0).reshape(3, -1)
    return ((kinv @ grid).reshape(3, H, W) * depth[None]).astype(np.float32)


def _normalize(v, eps=1e-12):
    n = np.sqrt((v * v).sum(0, keepdims=True, dtype=np.float32), dtype=np.float32)
    return v / np.maximum(n, np.float32(eps))


def surface_normal(P):
    """depth_post_proc.py:107-152 with nei = 1 on [3,H,W]."""
    c = P[:, 1:-1, 1:-1]
    sh = lambda dy, dx: P[:, 1 + dy:P.shape[1] - 1 + dy, 1 + dx:P.shape[2] - 1 + dx] - c
    x0, x1, y0, y1 = sh(0, -1), sh(0, 1), sh(-1, 0), sh(1, 0)
    x0y0, x0y1, x1y0, x1y1 = sh(-1, -1), sh(1, -1), sh(-1, 1), sh(1, 1)
    cross = lambda a, b: np.stack([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], 0)
    n = (_normalize(cross(x0, y0)) + _normalize(cross(x1, y1)) + _normalize(cross(x0y0, x0y1)) + _normalize(cross(x1y0, x1y1))) / np.float32(4)
    n = _normalize(n.astype(np.float32))
    return np.pad(n, ((0, 0), (1, 1), (1, 1)), mode="edge")


def ground_mask(P, n, threshold_deg=5):
    """depth_post_proc.py:155-185: |cos(normal, (0,1,0))| > cos(5 deg) and y > 0."""
    nn = np.sqrt((n * n).sum(0, dtype=np.float32), dtype=np.float32)
    cos = n[1] / np.maximum(nn * np.float32(1), np.float32(1e-6))          # CosineSimilarity(eps=1e-6) with a unit vector
    t = np.float32(np.cos(np.radians(threshold_deg)))
    return ((cos > t) | (cos < -t)) & ~(P[1] <= 0)


def depth_prediction(depth, use_dgc_scaling, K=None, real_camera_height=None, panoptic=None, road_class_id=-1,
                     depth_filter_class_ids=()):
    """depth [H,W] f32 -> (depth [H,W], xyz [3,H,W] or None, scale)."""
    d = np.asarray(depth, dtype=np.float32).copy()
    xyz, scale = None, np.float32(1)
    if use_dgc_scaling:
        xyz = reconstruct(d, np.asarray(K, dtype=np.float32))
        n = surface_normal(xyz)
        g = (panoptic == road_class_id) if panoptic is not None else ground_mask(xyz, n)
        h = np.abs((xyz * n).sum(0, dtype=np.float32))
        sel = np.sort(h[g])
        med = sel[(len(sel) - 1) // 2] if len(sel) else np.float32(np.nan)   # torch.median: the LOWER of the two middle values
        scale = np.float32(1) / med * np.float32(real_camera_height)      # :102 reciprocal().mul_()
        d *= scale
        xyz = xyz * scale
    if panoptic is not None:
        for cid in depth_filter_class_ids:
            d[panoptic == cid] = 0
            if xyz is not None:
                xyz[:, panoptic == cid] = np.nan
    return d, xyz, scale


def instance_predictions(sem_seg, center_heatmap, panoptic, thing_ids, label_divisor):
    """instance_post_proc.py:11-72 get_instance_predictions.  sem_seg [C,H,W] logits, center_heatmap [1,H,W] or [H,W], panoptic
    [H,W] -> (labels, classes, scores, boxes [n,4], masks [n,H,W] bool), segments in np.unique order (:38)."""
    sem = np.asarray(sem_seg, dtype=np.float32)
    heat = np.asarray(center_heatmap, dtype=np.float32).reshape(sem.shape[1:])
    pan = np.asarray(panoptic)
    z = sem - sem.max(0, keepdims=True)
    prob = np.exp(z, dtype=np.float32)
    prob = prob / prob.sum(0, keepdims=True, dtype=np.float32)                  # F.softmax(sem_seg, dim=0) (:36)
    labels, classes, scores, boxes, masks = [], [], [], [], []
    for lab in np.unique(pan):                                                    # :38
        if lab == -1:
            continue
        c = int(lab // label_divisor)                                             # :41
        if c not in thing_ids:                                                    # :42-44
            continue
        m = pan == lab                                                            # :50
        sem_score = np.float32(prob[c][m].mean(dtype=np.float64))                 # :53-54
        ys, xs = np.nonzero(m)                                                    # :56
        cy, cx = int(np.float32(ys.mean(dtype=np.float64))), int(np.float32(xs.mean(dtype=np.float64)))   # :57-61 (float32 means, int())
        scores.append(np.float32(sem_score * heat[cy, cx]))                       # :61-65
        boxes.append(np.array([xs.min(), ys.min(), xs.max() + 1, ys.max() + 1], np.float32))   # BitMasks.get_bounding_boxes (:67)
        labels.append(int(lab)); classes.append(c); masks.append(m)
    n = len(labels)
    return (np.array(labels, np.int64), np.array(classes, np.int64), np.array(scores, np.float32),
            np.stack(boxes) if n else np.zeros((0, 4), np.float32), np.stack(masks) if n else np.zeros((0,) + pan.shape, bool))


def pseudo_label_ids(panoptic, label_divisor, id_map):
    """tools/generate_pseudo_labels.py:100-118: panoptic prediction in train ids -> uint16 `instanceIds` image (the three masked
    assignments in the reference's order; id_map: uint8[256], trainId -> dataset id)."""
    p = np.asarray(panoptic).copy()
    id_map = np.asarray(id_map, dtype=np.uint8)
    sel = p % label_divisor == 0                                  # :103-106 stuff segments -> class train id
    p[sel] = p[sel] // label_divisor
    sel = p < label_divisor                                       # :107-109 train id -> id (numpy negative index for void = -1)
    p[sel] = id_map[p[sel]]
    sel = p >= label_divisor                                      # :110-118 things: id * divisor + instance
    # (uint8 * Python int: value-based promotion under the reference's NumPy < 2, exact for ids <= 65; int64 here)
    p[sel] = id_map[p[sel] // label_divisor].astype(np.int64) * label_divisor + p[sel] % label_divisor
    return p.astype(np.uint16)
