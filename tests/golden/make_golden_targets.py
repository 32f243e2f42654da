"""Generates tests/golden/targets_*.npz from the reference's OWN target generator (mgnet/data/target_generator.py, imported
unmodified by file path: it needs only numpy + torch).  Runs only where /root/reference exists (the build container):

    python tests/golden/make_golden_targets.py

Each fixture = inputs (label image, segment table, constructor arguments) + every entry of the dict the reference returns.
NumPy here is 2.x, so the offsets are the NEP-50 (float64) evaluation of target_generator.py:143-144 -- see
oracle/target_oracle.py.
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/mgnet/data/target_generator.py"

THING_IDS = list(range(11, 19))   # Cityscapes contiguous thing ids (mgnet/data/cityscapes_scene_seg.py)


def synth_case(seed, H, W, n_stuff=5, n_things=9, n_crowd=2, n_absent=2, unlabeled=True, small_blobs=3):
    """A label image with the features the generator reacts to: stuff bands, elliptical instances (some touching the
    border, some tiny, some overlapping so that earlier ones are partly or wholly hidden), crowd regions, unlabeled
    pixels (id 0, not in segments_info) and table rows whose segment is absent from the (cropped) image."""
    rng = np.random.default_rng(seed)
    pan = np.zeros((H, W), dtype=np.int32)
    segs = []
    yy, xx = np.mgrid[0:H, 0:W]
    bands = np.sort(rng.choice(np.arange(1, H - 1), size=n_stuff - 1, replace=False))
    edges = [0, *bands.tolist(), H]
    for k in range(n_stuff):
        cat = int(rng.integers(0, 11))
        sid = cat if k else 7   # stuff segments use the bare category id (cityscapes panoptic convention)
        if any(s["id"] == sid for s in segs):
            sid = 100 + k
        pan[edges[k]:edges[k + 1]] = sid
        segs.append(dict(id=sid, category_id=cat, iscrowd=0))
    def blob(cat, k, crowd, rmax):
        cy, cx = rng.uniform(-4, H + 4), rng.uniform(-4, W + 4)
        ry, rx = rng.uniform(1.0, rmax), rng.uniform(1.0, rmax * 1.6)
        m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        sid = cat * 1000 + k
        pan[m] = sid
        segs.append(dict(id=sid, category_id=cat, iscrowd=int(crowd)))
    for k in range(n_things):
        blob(int(rng.choice(THING_IDS)), k, False, min(H, W) / 4)
    for k in range(small_blobs):
        blob(int(rng.choice(THING_IDS)), 100 + k, False, 2.5)
    for k in range(n_crowd):
        blob(int(rng.choice(THING_IDS)), 200 + k, True, min(H, W) / 6)
    for k in range(n_absent):
        segs.append(dict(id=int(rng.choice(THING_IDS)) * 1000 + 300 + k, category_id=int(rng.choice(THING_IDS)), iscrowd=0))
    if unlabeled:
        m = ((yy - H * 0.8) / (H * 0.15)) ** 2 + ((xx - W * 0.5) / (W * 0.3)) ** 2 <= 1.0
        pan[m] = 0
    order = rng.permutation(len(segs))   # segments_info order is arbitrary (not sorted by id)
    return pan, [segs[i] for i in order]


CASES = {
    # name: (label-image arguments, generator arguments)
    "default": (dict(seed=1, H=96, W=160), dict(ignore_label=255, thing_ids=THING_IDS, sigma=8)),
    "cityscapes_cfg": (dict(seed=2, H=80, W=136), dict(ignore_label=255, thing_ids=THING_IDS, sigma=8, ignore_stuff_in_offset=True,
                                                        small_instance_area=4096, small_instance_weight=3)),
    "ragged_sigma3": (dict(seed=3, H=61, W=83), dict(ignore_label=255, thing_ids=THING_IDS, sigma=3, small_instance_area=40,
                                                      small_instance_weight=5, ignore_crowd_in_semantic=True)),
    "no_things": (dict(seed=4, H=40, W=64, n_things=0, n_crowd=0, n_absent=1, small_blobs=0), dict(ignore_label=255, thing_ids=THING_IDS, sigma=8)),
    "dense": (dict(seed=5, H=72, W=200, n_things=40, small_blobs=12, n_crowd=4), dict(ignore_label=255, thing_ids=THING_IDS, sigma=4,
                                                                                       ignore_stuff_in_offset=True, ignore_crowd_in_semantic=True)),
}


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_target_generator", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.PanopticDeepLabTargetGenerator


def seg_table(segs):
    return np.array([[s["id"], s["category_id"], s["iscrowd"]] for s in segs], dtype=np.int64).reshape(-1, 3)


def main():
    Gen = load_reference()
    for name, (img_kw, gen_kw) in CASES.items():
        pan, segs = synth_case(**img_kw)
        out = Gen(**gen_kw)(pan, segs)
        cp = np.array(out["center_points"], dtype=np.float64).reshape(-1, 2)
        sem = out["sem_seg"].numpy()
        assert sem.min() >= 0 and sem.max() <= 255
        for k in ("sem_seg_weights", "center_weights", "offset_weights"):
            v = out[k].numpy()
            assert np.array_equal(v, v.astype(np.uint8))
        np.savez_compressed(
            os.path.join(HERE, f"targets_{name}.npz"), panoptic=pan, segments=seg_table(segs),
            gen_keys=np.array(sorted(k for k in gen_kw if k != "thing_ids")), gen_vals=np.array([int(gen_kw[k]) for k in sorted(gen_kw) if k != "thing_ids"]),
            thing_ids=np.array(gen_kw["thing_ids"]), numpy_version=np.array(np.__version__),
            sem_seg=sem.astype(np.uint8), sem_seg_dtype=np.array(str(out["sem_seg"].dtype)), center=out["center"].numpy(), offset=out["offset"].numpy(),
            sem_seg_weights=out["sem_seg_weights"].numpy().astype(np.uint8), center_weights=out["center_weights"].numpy().astype(np.uint8),
            offset_weights=out["offset_weights"].numpy().astype(np.uint8), center_points=cp,
            shapes=np.array([list(out[k].shape) + [0] * (3 - out[k].dim()) for k in ("sem_seg", "center", "offset", "sem_seg_weights", "center_weights", "offset_weights")]))
        print(name, pan.shape, len(segs), "segments,", len(cp), "centres,", os.path.getsize(os.path.join(HERE, f"targets_{name}.npz")), "bytes")


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("reference not present: fixtures can only be regenerated in the build container")
    main()
