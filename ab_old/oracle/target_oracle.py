"""oracle/target_oracle.py -- numpy restatement of MGNet's panoptic target generation.

TEST INFRASTRUCTURE ONLY (never imported by mgnet_amd/).

Follows mgnet/data/target_generator.py:54-158 (PanopticDeepLabTargetGenerator.__call__), the Gaussian patch of :45-50,
`rgb2id` (panopticapi, used at mgnet/data/dataset_mapper.py:178: id = R + 256 G + 65536 B) and the class part of the
reprojection mask (dataset_mapper.py:214-216).

Parity status: **PINNED** by tests/golden/targets_*.npz = outputs of the reference's target_generator.py imported
unmodified in the build container (tests/golden/make_golden_targets.py; the file needs only numpy + torch).  The container
has NumPy 2.2, whose NEP-50 promotion evaluates `center_y - y_coord[mask]` (np.float64 scalar - float32 array, :143-144) in
float64; NumPy < 2 evaluates it in float32 (value-based casting).  The fixtures pin promotion="nep50"; "legacy" differs in
that one expression only and is what the reference computes on the NumPy versions it otherwise runs on (np.bool in
dataset_mapper.py:213 was removed in NumPy 1.24).

The restatement is organised per PIXEL (segment lookup -> table of per-segment results) instead of per segment, the way
the device kernels are, so it also documents their algorithm.
"""
import numpy as np


def gaussian_patch(sigma):
    """target_generator.py:45-50: (6 sigma + 3)^2 patch centred at 3 sigma + 1, float64."""
    r = np.arange(6 * sigma + 3, dtype=np.float64) - (3 * sigma + 1)
    return np.exp(-(r[None, :] ** 2 + r[:, None] ** 2) / (2 * sigma ** 2))


def rgb2id(rgb):
    """panopticapi.utils.rgb2id for uint8 [H,W,3]."""
    rgb = rgb.astype(np.int32)
    return rgb[..., 0] + 256 * rgb[..., 1] + 256 * 256 * rgb[..., 2]


def panoptic_targets(panoptic, segments_info, *, ignore_label, thing_ids, sigma=8, ignore_stuff_in_offset=False,
                     small_instance_area=0, small_instance_weight=1, ignore_crowd_in_semantic=False,
                     depth_ignore_ids=(), promotion="nep50"):
    """Returns numpy arrays keyed like the reference's dict (+ "reprojection_mask", "center_points")."""
    assert promotion in ("nep50", "legacy")
    thing_ids = sorted(thing_ids)
    H, W = panoptic.shape
    n = len(segments_info)
    ids = np.array([s["id"] for s in segments_info], dtype=np.int64).reshape(n)
    cat = np.array([s["category_id"] for s in segments_info], dtype=np.int64).reshape(n)
    crowd = np.array([bool(s["iscrowd"]) for s in segments_info], dtype=bool).reshape(n)
    thing = np.isin(cat, thing_ids)
    assert len(np.unique(ids)) == n, "segment ids must be unique"

    # pixel -> row of segments_info (or -1)
    flat = panoptic.reshape(-1).astype(np.int64)
    if n:
        order = np.argsort(ids)
        pos = np.clip(np.searchsorted(ids[order], flat), 0, n - 1)
        seg = np.where(ids[order][pos] == flat, order[pos], -1)
    else:
        seg = np.full(flat.shape, -1)
    hit = seg >= 0
    yy, xx = np.divmod(np.arange(H * W, dtype=np.int64), W)

    # per-segment area and centre (:105-121): np.mean over integer indices = exact integer sum / count in float64
    area = np.bincount(seg[hit], minlength=n).astype(np.int64)
    sum_y = np.zeros(n, dtype=np.int64)
    sum_x = np.zeros(n, dtype=np.int64)
    np.add.at(sum_y, seg[hit], yy[hit])
    np.add.at(sum_x, seg[hit], xx[hit])
    has_center = thing & ~crowd & (area > 0)
    with np.errstate(invalid="ignore", divide="ignore"):
        cy = np.where(has_center, sum_y.astype(np.float64) / area.astype(np.float64), np.nan)
        cx = np.where(has_center, sum_x.astype(np.float64) / area.astype(np.float64), np.nan)

    # per-segment values looked up per pixel; row n = "no segment"
    sem_of = np.where(ignore_crowd_in_semantic & crowd, ignore_label, cat)                       # :96-97
    inst_w = (~crowd & (thing | (not ignore_stuff_in_offset))).astype(np.float32)                 # :98-103
    small = has_center & (area < small_instance_area)                                             # :113-115
    semw_of = np.where(small, np.uint8(small_instance_weight & 255), np.uint8(1)).astype(np.float32)
    ext = lambda a, fill: np.concatenate([a, np.array([fill], dtype=a.dtype)])
    semantic = ext(sem_of.astype(np.int64), ignore_label)[seg].reshape(H, W)
    sem_w = ext(semw_of, np.float32(1))[seg].reshape(H, W)
    off_w = ext(inst_w, np.float32(0))[seg].reshape(H, W)
    ctr_w = np.where(semantic < thing_ids[0], np.float32(1), off_w).astype(np.float32)            # :146

    # offsets (:142-144)
    offset = np.zeros((2, H, W), dtype=np.float32)
    c_px = ext(has_center, False)[seg]
    cy_px, cx_px = ext(np.nan_to_num(cy), 0.0)[seg], ext(np.nan_to_num(cx), 0.0)[seg]
    if promotion == "legacy":
        oy = cy_px.astype(np.float32) - yy.astype(np.float32)
        ox = cx_px.astype(np.float32) - xx.astype(np.float32)
    else:
        oy = (cy_px - yy.astype(np.float64)).astype(np.float32)
        ox = (cx_px - xx.astype(np.float64)).astype(np.float32)
    offset[0] = np.where(c_px, oy, np.float32(0)).reshape(H, W)
    offset[1] = np.where(c_px, ox, np.float32(0)).reshape(H, W)

    # centre heat map (:117-139): maximum over the Gaussians anchored at the rounded centres, clipped to the frame
    g = gaussian_patch(sigma).astype(np.float32)   # max commutes with the (monotonic) rounding to float32
    G = 6 * sigma + 3
    center = np.zeros((H, W), dtype=np.float32)
    rows, cols = np.arange(H)[:, None], np.arange(W)[None, :]
    center_points = []
    for s in range(n):
        if not has_center[s]:
            continue
        center_points.append([cy[s], cx[s]])
        ul_y = int(np.round(cy[s])) - 3 * sigma - 1     # np.round: half to even
        ul_x = int(np.round(cx[s])) - 3 * sigma - 1
        y0, y1, x0, x1 = max(ul_y, 0), min(ul_y + G, H), max(ul_x, 0), min(ul_x + G, W)
        if y0 >= y1 or x0 >= x1:
            continue
        np.maximum(center[y0:y1, x0:x1], g[y0 - ul_y:y1 - ul_y, x0 - ul_x:x1 - ul_x], out=center[y0:y1, x0:x1])
    del rows, cols

    mask = ~np.isin(semantic, list(depth_ignore_ids))                                              # dataset_mapper.py:214-216
    return dict(sem_seg=semantic, center=center, center_points=center_points, offset=offset, sem_seg_weights=sem_w,
                center_weights=ctr_w[None], offset_weights=off_w[None], reprojection_mask=mask,
                seg_area=area)
