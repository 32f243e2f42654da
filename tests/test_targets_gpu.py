"""Device-side panoptic target generation (csrc/targets.hip through the C-ABI) against the reference's outputs
(tests/golden/targets_*.npz) and the numpy oracle.  Bit-exact: integer statistics, fp64 centres, gather-max heat map."""
import os

import numpy as np
import pytest
import torch

from conftest import TARGET_GOLD as GOLD
from conftest import golden_script
from conftest import load_target_case as load_case

from oracle import target_oracle as TO

_mk = golden_script("make_golden_targets")
THING_IDS, synth_case = _mk.THING_IDS, _mk.synth_case

pytestmark = pytest.mark.gpu

KEYS = ("sem_seg", "center", "offset", "sem_seg_weights", "center_weights", "offset_weights")


def gen(promotion="nep50", **kw):
    from mgnet_amd.data import PanopticDeepLabTargetGenerator
    return PanopticDeepLabTargetGenerator(promotion=promotion, **kw)


def assert_same(out, ref, b=None):
    for k in KEYS:
        got = out[k] if b is None else out[k][b]
        want = np.asarray(ref[k])
        assert tuple(got.shape) == want.shape, k
        assert np.array_equal(got.cpu().numpy(), want), k


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[8:-4] for p in GOLD])
def test_golden_reference_outputs(path):
    z, pan, segs, kw = load_case(path)
    out = gen(**kw)(pan, segs)
    assert out["sem_seg"].dtype == torch.int64 and out["center"].dtype == torch.float32
    assert np.array_equal(out["sem_seg"].cpu().numpy(), z["sem_seg"].astype(np.int64))
    assert np.array_equal(out["center"].cpu().numpy(), z["center"])
    assert np.array_equal(out["offset"].cpu().numpy(), z["offset"])
    for k in ("sem_seg_weights", "center_weights", "offset_weights"):
        assert tuple(out[k].shape) == tuple(int(s) for s in z["shapes"][KEYS.index(k)][:out[k].dim()])
        assert np.array_equal(out[k].cpu().numpy(), z[k].astype(np.float32)), k
    assert np.array_equal(np.array(out["center_points"], dtype=np.float64).reshape(-1, 2), z["center_points"])


@pytest.mark.parametrize("path", GOLD[:3], ids=[os.path.basename(p)[8:-4] for p in GOLD[:3]])
def test_legacy_promotion_matches_oracle(path):
    z, pan, segs, kw = load_case(path)
    out = gen("legacy", **kw)(pan, segs)
    assert_same(out, TO.panoptic_targets(pan, segs, promotion="legacy", **kw))


def test_rgb_labels_and_reprojection_mask():
    pan, segs = synth_case(seed=11, H=64, W=96)
    rgb = np.stack([pan & 255, (pan >> 8) & 255, (pan >> 16) & 255], -1).astype(np.uint8)
    assert np.array_equal(TO.rgb2id(rgb), pan)
    kw = dict(ignore_label=255, thing_ids=THING_IDS, sigma=8)
    g = gen(depth_ignore_ids=[10, 0, 13], **kw)
    a, b = g.generate_batch(pan[None], [segs]), g.generate_batch(rgb[None], [segs])
    ref = TO.panoptic_targets(pan, segs, depth_ignore_ids=[10, 0, 13], **kw)
    for out in (a, b):
        assert_same(out, ref, 0)
        assert out["reprojection_mask"].dtype == torch.bool
        assert np.array_equal(out["reprojection_mask"][0].cpu().numpy(), ref["reprojection_mask"])
    # ragged width (scalar path) with RGB labels
    pan, segs = synth_case(seed=12, H=37, W=51)
    rgb = np.stack([pan & 255, (pan >> 8) & 255, (pan >> 16) & 255], -1).astype(np.uint8)
    assert_same(g.generate_batch(rgb[None], [segs]), TO.panoptic_targets(pan, segs, **kw), 0)


def test_batch_with_different_tables():
    kw = dict(ignore_label=255, thing_ids=THING_IDS, sigma=5, small_instance_area=60, small_instance_weight=3)
    cases = [synth_case(seed=20, H=48, W=72), synth_case(seed=21, H=48, W=72, n_things=70, small_blobs=20),
             (np.zeros((48, 72), np.int32), [])]
    out = gen(**kw).generate_batch(np.stack([c[0] for c in cases]), [c[1] for c in cases], with_center_points=True)
    for b, (pan, segs) in enumerate(cases):
        ref = TO.panoptic_targets(pan, segs, **kw)
        assert_same(out, ref, b)
        assert out["center_points"][b] == [[float(y), float(x)] for y, x in ref["center_points"]]


def test_full_frame_vs_oracle_and_repeatable():
    pan, segs = synth_case(seed=30, H=1024, W=2048, n_stuff=9, n_things=90, small_blobs=40, n_crowd=6, n_absent=5)
    kw = dict(ignore_label=255, thing_ids=THING_IDS, sigma=8, small_instance_area=4096, small_instance_weight=3)
    g = gen(depth_ignore_ids=[10], **kw)
    dev_pan = torch.from_numpy(pan).cuda()
    a = g.generate_batch(dev_pan[None], [segs])
    ref = TO.panoptic_targets(pan, segs, depth_ignore_ids=[10], **kw)
    assert_same(a, ref, 0)
    assert np.array_equal(a["reprojection_mask"][0].cpu().numpy(), ref["reprojection_mask"])
    b = g.generate_batch(torch.stack([dev_pan, dev_pan.flip(0)]), [segs, segs])   # other launch shape, same image in slot 0
    for k in KEYS:
        assert torch.equal(a[k][0], b[k][0]), k
    # flipping the label image flips every map; vertical offsets change sign
    assert torch.equal(b["sem_seg"][1].flip(0), b["sem_seg"][0])
    assert torch.equal(b["offset"][1, 1].flip(0), b["offset"][0, 1])


def test_errors_are_loud():
    from mgnet_amd.data import PanopticDeepLabTargetGenerator
    kw = dict(ignore_label=255, thing_ids=THING_IDS)
    pan = np.zeros((8, 8), np.int32)
    with pytest.raises(ValueError):
        gen(**kw)(pan, [dict(id=1, category_id=1, iscrowd=0), dict(id=1, category_id=2, iscrowd=0)])
    with pytest.raises(ValueError):
        gen(**kw)(pan, [dict(id=i + 1, category_id=1, iscrowd=0) for i in range(1025)])
    with pytest.raises(RuntimeError):
        PanopticDeepLabTargetGenerator(device="cpu", **kw)(pan, [])
    with pytest.raises(IndexError):
        gen(ignore_label=255, thing_ids=[])(pan, [])


def test_generated_targets_feed_the_training_step(torch_staging):
    """Label images -> device target maps -> MGNet.forward: the per-image dicts hold slices of the batched maps, which the
    batch assembly takes without a copy; losses equal those with the oracle's (host-generated) maps."""
    from test_network_cpu import small_model

    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.structures import ImageList

    H, W, B = 64, 96, 2
    kw = dict(ignore_label=255, thing_ids=THING_IDS, sigma=8, small_instance_area=200, small_instance_weight=3)
    cases = [synth_case(seed=40 + b, H=H, W=W) for b in range(B)]
    out = gen(depth_ignore_ids=[10], **kw).generate_batch(np.stack([c[0] for c in cases]), [c[1] for c in cases])
    assert ImageList.from_tensors([out["offset"][b] for b in range(B)], 32).tensor.data_ptr() == out["offset"].data_ptr()

    cfg, m = small_model(with_depth=True, seed=3)
    m = m.cuda().train()
    batch = synthetic_batch(B, H, W, "cuda", seed=5)
    dev_batch, ref_batch = [], []
    for b in range(B):
        ref = TO.panoptic_targets(cases[b][0], cases[b][1], depth_ignore_ids=[10], **kw)
        d, r = dict(batch[b]), dict(batch[b])
        for k in KEYS + ("reprojection_mask",):
            d[k] = out[k][b]
            r[k] = torch.from_numpy(np.ascontiguousarray(ref[k])).cuda()
        dev_batch.append(d)
        ref_batch.append(r)
    torch.manual_seed(0)
    got = m(dev_batch)
    want = m(ref_batch)
    for k in want:   # (the target maps are bit-identical; the loss reductions use float atomics, so not bit-repeatable)
        assert float(got[k].detach()) == pytest.approx(float(want[k].detach()), rel=1e-5), k
