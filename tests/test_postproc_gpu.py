"""Inference post-processing on the device (csrc/postproc.hip through the C-ABI) against the reference's outputs
(tests/golden/postproc_*.npz) and the numpy oracle.  Panoptic ids: bit-exact (integer work, first-minimum/-maximum rules)."""
import glob
import os

import numpy as np
import pytest
import torch
from conftest import GOLDEN, golden_script

from oracle import postproc_oracle as PO

pytestmark = pytest.mark.gpu

PAN = sorted(glob.glob(os.path.join(GOLDEN, "postproc_pan_*.npz")))
MK = golden_script("make_golden_postproc")


def pan_kwargs(z):
    kw = {str(k): float(v) for k, v in zip(z["kw_keys"], z["kw_vals"])}
    return dict(MK.PAN_KW, stuff_area=int(kw["stuff_area"]), threshold=kw["threshold"], nms_kernel=int(kw["nms_kernel"]))


def run(sem, center, off, **kw):
    from mgnet_amd.postprocessing import get_panoptic_prediction
    s, c, o = torch.from_numpy(sem.astype(np.int64))[None].cuda(), torch.from_numpy(center)[None].cuda(), torch.from_numpy(off).cuda()
    keep = (s.clone(), c.clone(), o.clone())
    out = get_panoptic_prediction(s, c, o, **kw)
    assert all(torch.equal(a, b) for a, b in zip(keep, (s, c, o))), "inputs must stay untouched"
    assert out.dtype == torch.int64 and tuple(out.shape) == sem.shape
    return out.cpu().numpy()


@pytest.mark.parametrize("path", PAN, ids=[os.path.basename(p)[13:-4] for p in PAN])
def test_golden_reference_outputs(path):
    z = np.load(path)
    out = run(z["sem"], z["center"], z["offsets"], **pan_kwargs(z))
    assert np.array_equal(out, z["panoptic"].astype(np.int64))


@pytest.mark.parametrize("shape,n_inst", [((256, 512), 40), ((130, 2100), 25), ((1024, 2048), 150)])
def test_larger_frames_vs_oracle(shape, n_inst):
    H, W = shape
    sem, center, off = MK.pan_case(seed=H + n_inst, H=H, W=W, n_inst=n_inst, noise=2.0, plateau=True)
    kw = dict(MK.PAN_KW, stuff_area=2048, threshold=0.3, nms_kernel=7)
    ref = PO.panoptic_prediction(sem, center, off, **kw)
    out = run(sem, center, off, **kw)
    assert np.array_equal(out, ref)
    assert (out >= 11000).any()


def test_many_centres_beyond_one_lds_chunk():
    """> 2048 centres (several LDS chunks): a noisy heat map with a 3x3 NMS."""
    rs = np.random.RandomState(0)
    H, W = 192, 320
    sem = rs.randint(0, 19, size=(H, W)).astype(np.int64)
    center = rs.rand(H, W).astype(np.float32)
    off = (rs.randn(2, H, W) * 20).astype(np.float32)
    kw = dict(MK.PAN_KW, stuff_area=10, threshold=0.3, nms_kernel=3)
    assert len(PO.center_points(center, 0.3, 3)) > 2048
    assert np.array_equal(run(sem, center, off, **kw), PO.panoptic_prediction(sem, center, off, **kw))


def test_argument_checks_and_overflow():
    from mgnet_amd.postprocessing import get_panoptic_prediction
    kw = dict(MK.PAN_KW, stuff_area=1)
    s, c, o = torch.zeros(1, 8, 8, dtype=torch.int64).cuda(), torch.zeros(1, 8, 8).cuda(), torch.zeros(2, 8, 8).cuda()
    with pytest.raises(ValueError):
        get_panoptic_prediction(s, c[0], o, **kw)
    with pytest.raises(ValueError):
        get_panoptic_prediction(s, c, o[0], **kw)
    with pytest.raises(RuntimeError):
        get_panoptic_prediction(s.cpu(), c.cpu(), o.cpu(), **kw)
    # a constant heat map above the threshold: every pixel is a (tied) maximum -> more centres than the id space
    H, W = 300, 300
    with pytest.raises(RuntimeError, match="centre points"):
        get_panoptic_prediction(torch.full((1, H, W), 12, dtype=torch.int64).cuda(), torch.full((1, H, W), 0.5).cuda(),
                                torch.zeros(2, H, W).cuda(), **kw)


# ---- depth ----------------------------------------------------------------------------------------------------------------
DEP = sorted(glob.glob(os.path.join(GOLDEN, "postproc_depth_*.npz")))


def run_depth(depth, use_dgc, K, pan, **kw):
    from mgnet_amd.postprocessing import get_depth_prediction
    d_in = torch.from_numpy(depth)[None, None].cuda()
    keep = d_in.clone()
    d, xyz = get_depth_prediction(d_in, use_dgc, camera_matrix=torch.from_numpy(K)[None], real_camera_height=torch.tensor([1.65]),
                                  panoptic_seg=None if pan is None else torch.from_numpy(pan.astype(np.int64)).cuda(), **kw)
    assert torch.equal(keep, d_in)
    return d.cpu().numpy(), None if xyz is None else xyz.cpu().numpy()


def check_depth(d, xyz, d_ref, xyz_ref, rtol=3e-6):
    assert np.array_equal(d == 0, d_ref == 0)
    np.testing.assert_allclose(d, d_ref, rtol=rtol, atol=0)
    if xyz_ref is None:
        assert xyz is None
    else:
        assert np.array_equal(np.isnan(xyz), np.isnan(xyz_ref))
        np.testing.assert_allclose(np.nan_to_num(xyz), np.nan_to_num(xyz_ref), rtol=rtol, atol=2e-6)


@pytest.mark.parametrize("path", DEP, ids=[os.path.basename(p)[15:-4] for p in DEP])
def test_depth_golden_reference_outputs(path):
    z = np.load(path)
    pan = z["panoptic"] if bool(z["with_pan"]) else None
    d, xyz = run_depth(z["depth_in"], bool(z["use_dgc"]), z["K"], pan, **(MK.DEPTH_KW if pan is not None else {}))
    check_depth(d, xyz, z["depth"], z["xyz"] if bool(z["use_dgc"]) else None)


@pytest.mark.parametrize("with_pan", [True, False])
def test_depth_full_frame_vs_oracle(with_pan):
    depth, K, pan = MK.depth_case(5, 512, 1024)
    pan = pan if with_pan else None
    kw = MK.DEPTH_KW if with_pan else {}
    d_ref, xyz_ref, scale = PO.depth_prediction(depth, True, K=K, real_camera_height=1.65, panoptic=pan, **kw)
    d, xyz = run_depth(depth, True, K, pan, **kw)
    # the normals come from differences of neighbouring points whose x/y are ~W/2 times larger than the differences: fp32
    # round-off of the (equally valid) evaluation orders is amplified by ~fx, so the tolerance scales with the frame width
    check_depth(d, xyz, d_ref, xyz_ref, rtol=1e-4)
    assert np.isfinite(float(scale))


def test_depth_argument_checks():
    from mgnet_amd.postprocessing import get_depth_prediction
    d = torch.ones(1, 1, 8, 8).cuda()
    with pytest.raises(AssertionError):
        get_depth_prediction(d, True)
    with pytest.raises(AssertionError):
        get_depth_prediction(d, True, camera_matrix=torch.eye(3)[None], real_camera_height=torch.tensor([1.0]),
                             panoptic_seg=torch.zeros(8, 8, dtype=torch.int64).cuda())
    with pytest.raises(RuntimeError):
        get_depth_prediction(d.cpu(), False)
    out, xyz = get_depth_prediction(d, False, panoptic_seg=torch.zeros(8, 8, dtype=torch.int64).cuda(), depth_filter_class_ids=[0])
    assert xyz is None and float(out.abs().sum()) == 0.0


# ---- the inference branch of MGNet.forward (mg_net.py:375-425) ----------------------------------------------------------
def test_eval_forward_runs_the_post_processing(torch_staging):
    """model.eval()(batch) -> per-image dicts; the panoptic ids / metric depth equal the oracle's post-processing of the
    very head outputs the model produced (checks the wiring: argmax, config values, road mask, filtered classes)."""
    from test_network_cpu import make_cfg

    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.modeling import mg_net as MG
    from mgnet_amd.registry import build_model

    torch.manual_seed(0)
    cfg = make_cfg(os.path.join(os.path.dirname(os.path.dirname(__file__)), "configs", "bench-c4-cityscapes-videosequence.yaml"),
                   **{"MODEL.DEVICE": "cpu", "SOLVER.AMP.ENABLED": False, "MODEL.POST_PROCESSING.STUFF_AREA": 40,
                      "INPUT.IGNORED_CATEGORIES_IN_DEPTH": ["sky", "ego vehicle"]})
    m = build_model(cfg).cuda().eval()
    with torch.no_grad():   # make the random heads produce a varied picture
        w = m.sem_seg_head.head.predictor.weight
        w.normal_(0, 0.5)
        m.ins_embed_head.offset_head.predictor.weight.normal_(0, 0.05)
        m.ins_embed_head.center_head.predictor.weight.normal_(0, 0.05)
    H, W = 64, 96
    batch = synthetic_batch(2, H, W, "cuda", seed=3)
    for d in batch:
        d["camera_matrix"] = d["camera_matrix"][:3, :3].cpu()
        d["camera_height"] = torch.tensor([1.22])
    with torch.no_grad():   # make "road" (train id 1) the most frequent stuff class of this random network
        sem0 = torch.stack([r["sem_seg"].argmax(0) for r in m(batch)])
        top = int(torch.bincount(sem0.flatten(), minlength=20)[:12].argmax())
        w = m.sem_seg_head.head.predictor.weight
        w[[1, top]] = w[[top, 1]].clone()
    seen = {}
    orig = MG.MGNet._inference

    def spy(self, bi, outputs):
        seen.update({k: v.detach().float().cpu().numpy() for k, v in outputs.items()})
        return orig(self, bi, outputs)
    MG.MGNet._inference = spy
    try:
        with torch.no_grad():
            res = m(batch)
    finally:
        MG.MGNet._inference = orig
    assert len(res) == 2 and set(res[0]) == {"sem_seg", "panoptic_seg", "depth"}
    K = batch[0]["camera_matrix"].numpy()
    for b, r in enumerate(res):
        pan, none = r["panoptic_seg"]
        assert none is None and pan.dtype == torch.int64 and tuple(pan.shape) == (H, W)
        assert tuple(r["sem_seg"].shape) == (20, H, W)
        sem = seen["sem_seg"][b].argmax(0)
        want = PO.panoptic_prediction(sem, seen["center"][b, 0], seen["offset"][b], num_thing_classes=8, last_stuff_id=11,
                                      label_divisor=1000, stuff_area=40, void_label=-1, threshold=0.3, nms_kernel=7)
        assert np.array_equal(pan.cpu().numpy(), want)
        depth, xyz = r["depth"]
        d_ref, xyz_ref, scale = PO.depth_prediction(seen["depth"][b, 0], True, K=K, real_camera_height=1.22, panoptic=want,
                                                    road_class_id=1000, depth_filter_class_ids=[0, 11000])
        got = depth.cpu().numpy()
        assert (want == 1000).sum() > 50 and np.isfinite(scale)
        assert np.array_equal(got == 0, d_ref == 0)
        np.testing.assert_allclose(got, d_ref, rtol=1e-4)
        assert tuple(xyz.shape) == (3, H, W) and np.array_equal(np.isnan(xyz.cpu().numpy()), np.isnan(xyz_ref))


INS = sorted(glob.glob(os.path.join(GOLDEN, "instances_*.npz")))


@pytest.mark.parametrize("path", INS, ids=[os.path.basename(p)[10:-4] for p in INS])
def test_instance_predictions_match_reference(path):
    """mgnet_amd.postprocessing.get_instance_predictions (csrc/instances.hip) against the reference's own outputs: classes, boxes
    and masks exactly, scores to fp32 round-off (the mean class probability is an order-independent fixed-point sum here)."""
    from mgnet_amd.postprocessing import get_instance_predictions
    from mgnet_amd.structures import Instances

    z = np.load(path)
    sem, heat, pan = (torch.from_numpy(z[k]).cuda() for k in ("sem", "heat", "pan"))
    out = get_instance_predictions(sem, heat, pan, list(z["thing_ids"]), int(z["label_divisor"]))
    n = len(z["classes"])
    assert len(out) == n
    if n == 0:
        return
    cat = Instances.cat(out)
    assert cat.image_size == tuple(pan.shape) and len(cat) == n
    assert np.array_equal(cat.pred_classes.cpu().numpy(), z["classes"])
    assert np.array_equal(cat.pred_boxes.tensor.cpu().numpy(), z["boxes"])
    np.testing.assert_allclose(cat.scores.cpu().numpy(), z["scores"], rtol=1e-5, atol=1e-8)
    masks = cat.pred_masks.cpu().numpy()
    assert masks.dtype == np.bool_ and np.array_equal(np.packbits(masks.reshape(n, -1), axis=1), z["masks"])


def test_instance_predictions_large_frame_vs_oracle():
    """1024x2048 frame with ~200 thing segments against the numpy oracle; result independent of repetition"""
    from mgnet_amd import _C
    from oracle import postproc_oracle as PO

    rs = np.random.RandomState(0)
    H, W, C = 1024, 2048, 19
    blk = rs.randint(0, 11, size=(H // 16, W // 16))
    pan = (np.kron(blk, np.ones((16, 16), dtype=np.int64)) * 1000).astype(np.int64)
    yy, xx = np.mgrid[0:H, 0:W]
    for k in range(200):
        cy, cx, r = rs.randint(H), rs.randint(W), rs.randint(3, 90)
        pan[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = rs.randint(11, 19) * 1000 + k + 1
    sem = rs.randn(C, H, W).astype(np.float32)
    heat = rs.rand(H, W).astype(np.float32)
    ref = PO.instance_predictions(sem, heat, pan, list(range(11, 19)), 1000)
    args = (torch.from_numpy(sem).cuda(), torch.from_numpy(heat).cuda(), torch.from_numpy(pan).cuda(), list(range(11, 19)), 1000)
    a = _C.instance_post(*args)
    b = _C.instance_post(*args)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert np.array_equal(a[0].cpu().numpy(), ref[0]) and np.array_equal(a[1].cpu().numpy(), ref[1])
    assert np.array_equal(a[3].cpu().numpy(), ref[3])
    np.testing.assert_allclose(a[2].cpu().numpy(), ref[2], rtol=2e-5)
    assert np.array_equal(a[4].cpu().numpy(), ref[4])


def test_pseudo_label_ids_match_reference():
    """mgn_pseudo_label_ids against the reference's own arithmetic (fixture) and the oracle on a full frame"""
    from mgnet_amd import _C

    z = np.load(os.path.join(GOLDEN, "pseudo_labels.npz"))
    out = _C.pseudo_label_ids(torch.from_numpy(z["pan"]).cuda(), int(z["label_divisor"]), z["id_map"])
    assert np.array_equal(out.cpu().numpy().view(np.uint16), z["out"])
    rs = np.random.RandomState(1)
    pan = (rs.randint(0, 19, size=(1024, 2048)) * 1000 + rs.randint(0, 40, size=(1024, 2048)) * (rs.rand(1024, 2048) < 0.5)).astype(np.int64)
    pan[rs.rand(1024, 2048) < 0.02] = -1
    ref = PO.pseudo_label_ids(pan, 1000, z["id_map"])
    got = _C.pseudo_label_ids(torch.from_numpy(pan).cuda(), 1000, z["id_map"]).cpu().numpy().view(np.uint16)
    assert np.array_equal(got, ref)


def test_eval_forward_with_instances():
    """TEST.EVAL_INSTANCE (mg_net.py:145-153, 394-402): `model.eval()(batch)` carries an `instances` entry that is consistent with
    its own panoptic prediction (one instance per thing segment in ascending id order, class = id // divisor, mask = segment,
    tight box) -- the scores are pinned by the fixtures above."""
    from mgnet_amd import add_mgnet_config, get_cfg
    from mgnet_amd.registry import build_model
    from conftest import ROOT

    cfg = get_cfg()
    add_mgnet_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "bench-c4-cityscapes-videosequence.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cuda:0", "WITH_DEPTH", False, "TEST.EVAL_INSTANCE", True, "MODEL.POST_PROCESSING.CENTER_THRESHOLD", 0.0001])
    torch.manual_seed(1)
    model = build_model(cfg).eval()
    g = torch.Generator().manual_seed(2)
    batch = [{"image": torch.randint(0, 256, (3, 128, 256), generator=g, dtype=torch.uint8).cuda()} for _ in range(2)]
    with torch.no_grad():
        out = model(batch)
    n_total = 0
    for r in out:
        pan = r["panoptic_seg"][0]
        ids = [int(v) for v in torch.unique(pan).tolist() if v != -1 and 12 <= v // 1000 <= 19]   # Cityscapes thing train ids (20 classes)
        if not ids:
            assert "instances" not in r
            continue
        ins = r["instances"]
        assert len(ins) == len(ids) and ins.image_size == tuple(pan.shape)
        assert ins.pred_classes.tolist() == [v // 1000 for v in ids]
        for k, v in enumerate(ids):
            m = pan == v
            assert torch.equal(ins.pred_masks[k], m)
            ys, xs = torch.nonzero(m, as_tuple=True)
            assert ins.pred_boxes.tensor[k].tolist() == [float(xs.min()), float(ys.min()), float(xs.max() + 1), float(ys.max() + 1)]
        assert torch.isfinite(ins.scores).all() and (ins.scores >= 0).all()
        n_total += len(ids)
    assert n_total > 0, "the random model produced no thing segment: lower the centre threshold of this test"
