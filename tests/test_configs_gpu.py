"""GPU: the other BASELINE.json configurations through the HIP path against the CPU oracle -- C3 (KITTI-shaped 192x640
frames, depth + reprojection) and C2 (panoptic only, WITH_DEPTH False), bf16 like the benchmark.  Small batches so that the
oracle finishes in seconds; these shapes exercise narrow feature maps (6x20 at stride 32) and the panoptic-only graph."""
import pytest
import torch
from test_network_cpu import small_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,with_panoptic,with_depth,H,W", [("C3-kitti-192x640", True, True, 192, 640),
                                                               ("C3-depth-only", False, True, 192, 640),
                                                               ("C2-panoptic-only", True, False, 256, 512)])
def test_config_shapes_match_oracle(name, with_panoptic, with_depth, H, W):
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer
    from oracle import network_oracle as NO

    cfg, m = small_model(with_depth=with_depth, with_panoptic=with_panoptic, seed=4)
    m.train()
    batch = synthetic_batch(2, H, W, "cpu", seed=6, with_panoptic=with_panoptic, with_depth=with_depth)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ref = NO.mgnet_losses(sd, batch, pixel_mean=cfg.MODEL.PIXEL_MEAN, pixel_std=cfg.MODEL.PIXEL_STD, ohem_n_min=1500,
                          with_panoptic=with_panoptic, with_depth=with_depth)
    m = m.cuda()
    m.amp_dtype = torch.bfloat16
    dev_batch = [{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in x.items()} for x in batch]
    got = m(dev_batch)
    want_keys = (["loss_sem_seg", "loss_center", "loss_offset"] if with_panoptic else []) + \
                (["loss_photometric", "loss_smoothness"] if with_depth else [])
    assert list(got) == want_keys == list(ref)
    for k in ref:
        assert float(got[k].detach()) == pytest.approx(float(ref[k]), rel=4e-2, abs=2e-4), (name, k)
    # and a few optimizer steps run (fused Adam on the flat buckets) with finite, decreasing total loss
    tr = Trainer(cfg, m)
    tot = [float(sum(v.detach() for v in tr.run_step(dev_batch).values())) for _ in range(6)]
    assert all(t == t and abs(t) < 1e6 for t in tot) and tot[-1] < tot[0], tot


def test_full_size_step_properties():
    """BASELINE C4 frame size (1024 x 2048, bf16, 4 frames): the oracle cannot run this in seconds, so size-independent
    properties instead -- finite losses and gradients for every parameter, the five weighted losses in the reference's order,
    and invariance of the (per-rank mean) losses under a permutation of the frames inside the batch."""
    import os

    from mgnet_amd import add_mgnet_config, get_cfg
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.registry import build_model

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = get_cfg()
    add_mgnet_config(cfg)
    cfg.merge_from_file(os.path.join(root, "configs", "bench-c4-cityscapes-videosequence.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cuda", "MODEL.SEM_SEG_HEAD.OHEM_N_MIN", 524287])
    torch.manual_seed(0)
    m = build_model(cfg).train()
    batch = synthetic_batch(4, 1024, 2048, "cuda", seed=21)
    got = m(batch)
    assert list(got) == ["loss_sem_seg", "loss_center", "loss_offset", "loss_photometric", "loss_smoothness"]
    sum(got.values()).backward()
    vals = {k: float(v.detach()) for k, v in got.items()}
    assert all(v == v and abs(v) < 1e4 for v in vals.values()), vals
    for n, p in m.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
    perm = m([batch[2], batch[0], batch[3], batch[1]])
    for k, v in vals.items():   # batch statistics and masked means are permutation invariant; bf16 + atomics: loose tolerance
        assert float(perm[k].detach()) == pytest.approx(v, rel=2e-2, abs=2e-4), k


def _c4_trainer(B=8):
    import os

    from mgnet_amd import add_mgnet_config, get_cfg
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer
    from mgnet_amd.registry import build_model

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = get_cfg()
    add_mgnet_config(cfg)
    cfg.merge_from_file(os.path.join(root, "configs", "bench-c4-cityscapes-videosequence.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cuda", "SOLVER.IMS_PER_BATCH", B])
    torch.manual_seed(0)
    return Trainer(cfg, build_model(cfg)), synthetic_batch(B, 1024, 2048, torch.device("cuda"), seed=21)


def test_full_size_eager_steps_are_reproducible():
    """BASELINE C4 / C5 at the benchmark's own size (8 frames of 1024 x 2048, bf16): training steps run three times from one state snapshot.
    Round 6 found that they did NOT agree: the windowed 3x3 kernels read their last weight stages before they had landed whenever the
    memory system was busy (csrc/conv_win.hip, compiler-merged dummy loads under a constant vmcnt wait) -- invisible at the small shapes
    of the other tests, ~1 % wrong tiles here, all five losses of the FIRST step off by 1e-5 .. 1e-4 relative in every repetition.
    The bound asserted on the first step is 2e-6 relative, not bit-equality: with the kernels fixed (tests/test_race_gpu.py holds them to
    the bit, one at a time) a full-size step on this platform still differs in the last bit of ONE loss in roughly one of ten steps --
    a kernel's output differs while checksums of its inputs taken right before and right after it agree, the signature of a stale cache
    line from the previous step (profiles/r06_determinism.txt section 4); later steps amplify it chaotically and are only required to
    stay close and finite."""
    tr, batch = _c4_trainer()
    for _ in range(2):
        tr.run_step(batch)
    snap = tr.state_snapshot()
    runs = []
    for _ in range(3):
        tr.state_restore(snap)
        losses = []
        for _ in range(3):
            ld = tr.run_step(batch)   # (no synchronisation between the steps: the host runs ahead as in a training loop)
            losses.append(torch.stack([v.detach().float() for v in ld.values()]).clone())
        torch.cuda.synchronize()
        runs.append(torch.stack(losses).double().cpu())
    ref = runs[0]
    exact = sum(bool(torch.equal(ref, r)) for r in runs[1:])
    print(f"[full-size eager steps] {exact} of {len(runs) - 1} repetitions bit-identical to the first over 3 steps")
    for k, r in enumerate(runs[1:], 1):
        rel = ((r - ref).abs() / ref.abs().clamp_min(1e-12))
        assert float(rel[0].max()) <= 2e-6, (k, "first step", ref[0].tolist(), r[0].tolist())
        assert float(rel.max()) <= 2e-3, (k, ref.tolist(), r.tolist())
    assert bool(torch.isfinite(ref).all()) and float(ref[-1].sum()) < float(ref[0].sum())


def test_full_size_recordings_are_checked_before_they_are_used():
    """The launch-plan replay at the benchmark's size goes through Trainer.record_plan(best_of, verify_steps) (bench.py does this before
    its timed region): two recordings replayed from one state must agree bit for bit, else the read-only declarations are
    dropped and the step is recorded and checked again, else the recording is refused and the trainer stays on the eager step.  Whichever
    way it ends, the outcome is reported in `plan_check` and the trainer keeps training.  (Round 6, MI355X: the replays of the full-size
    step differ in the last bits of one loss in a few per cent of the steps, eager steps included -- profiles/r06_determinism.txt -- so the refusal
    branch is the one usually taken here; at the small shapes of tests/test_plan_gpu.py the check passes.)"""
    tr, batch = _c4_trainer()
    for _ in range(3):
        tr.run_step(batch)
    from mgnet_amd.engine import plan as plan_mod
    try:
        try:
            tr.record_plan(batch, best_of=2, trial_steps=2, verify_steps=8)
            accepted = True
        except RuntimeError as e:
            assert "replay to different results" in str(e), e
            accepted = False
        chk = tr.plan_check
        assert chk is not None and chk["steps"] == 8 and "first_difference" in chk
        print(f"[full-size plan check] accepted={accepted} {chk}")
        if accepted:
            assert chk["identical"] and tr._plan is not None
            last = {k: float(v) for k, v in tr.replay_plan().items()}
        else:
            assert not chk["identical"] and tr._plan is None
            last = {k: float(v) for k, v in tr.run_step(batch).items()}   # the eager step is what remains, and it works
        assert all(v == v and abs(v) < 1e4 for v in last.values()), last
    finally:
        plan_mod.set_ro_mode(None)
        if getattr(tr, "_plan", None) is not None:
            tr._plan.close()
