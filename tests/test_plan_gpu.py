"""Launch-plan replay (engine/plan.py, csrc/plan.hip): a step recorded once and replayed from C gives the SAME trajectory, bit for bit,
as eager steps -- losses and every parameter after several optimizer steps, side streams on, refilled inputs seen."""
import copy
import os

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _trainer(seed=0, H=128, W=256, B=2, dtype="bfloat16"):
    from mgnet_amd import add_mgnet_config, get_cfg
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer
    from mgnet_amd.registry import build_model

    dev = torch.device("cuda:0")
    cfg = get_cfg()
    add_mgnet_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "bench-c4-cityscapes-videosequence.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", str(dev), "SOLVER.IMS_PER_BATCH", B, "MODEL.SEM_SEG_HEAD.OHEM_N_MIN", B * H * W // 16,
                         "SOLVER.AMP.DTYPE", dtype])
    torch.manual_seed(seed)
    model = build_model(cfg)
    return Trainer(cfg, model), synthetic_batch(B, H, W, dev, seed=11), synthetic_batch(B, H, W, dev, seed=12)


def _refill(dst, src):
    for d, s in zip(dst, src):
        for k, v in d.items():
            if isinstance(v, torch.Tensor):
                v.copy_(s[k])


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_replayed_steps_equal_eager_steps_bit_for_bit(dtype):
    steps = 4
    ta, batch_a, other_a = _trainer(dtype=dtype)
    tb, batch_b, other_b = _trainer(dtype=dtype)
    # reference: eager steps (3 warm-up + `steps`, the input changes half way)
    la = []
    for k in range(3 + steps):
        if k == 3 + steps // 2:
            _refill(batch_a, other_a)
        la.append({n: float(v) for n, v in ta.run_step(batch_a).items()})
    # plan: 3 eager warm-up steps, one recorded step, then replays
    lb = []
    for k in range(3):
        lb.append({n: float(v) for n, v in tb.run_step(batch_b).items()})
    plan = tb.record_plan(batch_b)
    lb.append({n: float(v) for n, v in tb._plan_losses.items()})
    for k in range(4, 3 + steps):
        if k == 3 + steps // 2:
            _refill(batch_b, other_b)
        lb.append({n: float(v) for n, v in tb.replay_plan().items()})
    torch.cuda.synchronize()
    assert plan.report["kernel_launches"] > 300 and plan.report["streams"] >= 2, plan.report
    for k, (a, b) in enumerate(zip(la, lb)):
        assert a == b, (k, a, b, plan.report)
    for (na, pa), (nb, pb) in zip(ta.model.named_parameters(), tb.model.named_parameters()):
        assert torch.equal(pa, pb), na
    for (na, ba), (nb, bb) in zip(ta.model.named_buffers(), tb.model.named_buffers()):
        assert torch.equal(ba, bb), na


def test_replay_is_repeatable_and_coexists_with_eager_work():
    """eager work between replays (another model's steps: allocator traffic, library counters) does not disturb the plan's memory"""
    ta, batch_a, _ = _trainer(seed=1)
    for _ in range(3):
        ta.run_step(batch_a)
    ta.record_plan(batch_a)
    tb, batch_b, _ = _trainer(seed=2)
    ref, _, _ = _trainer(seed=1)
    ref_batch = batch_a
    for _ in range(4):
        ref.run_step(ref_batch)
    for k in range(3):
        tb.run_step(batch_b)
        la = {n: float(v) for n, v in ta.replay_plan().items()}
        lr = {n: float(v) for n, v in ref.run_step(ref_batch).items()}
        assert la == lr, (k, la, lr)


def test_run_step_planned_is_run_step_with_a_new_batch_every_call():
    """the training-loop entry: eager warm-up, one recorded step, then refill + replay with a DIFFERENT batch per call -- the trajectory of
    run_step on the same batches, bit for bit; a batch of another shape takes the eager step and the plan survives it"""
    from mgnet_amd.data import synthetic_batch
    ta, _, _ = _trainer()
    tb, _, _ = _trainer()
    dev = torch.device("cuda:0")
    batches = [synthetic_batch(2, 128, 256, dev, seed=20 + k) for k in range(7)]
    batches[5] = [{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in d.items()} for d in batches[5]]   # (not collated)
    la = [{n: float(v) for n, v in ta.run_step(b).items()} for b in batches]
    lb = [{n: float(v) for n, v in tb.run_step_planned(b).items()} for b in batches]
    assert tb._plan is not None and getattr(tb, "plan_note", None) is None and getattr(tb, "plan_eager_steps", 0) == 0
    assert la == lb
    odd = synthetic_batch(2, 128, 192, dev, seed=40)
    la.append({n: float(v) for n, v in ta.run_step(odd).items()})
    lb.append({n: float(v) for n, v in tb.run_step_planned(odd).items()})
    la.append({n: float(v) for n, v in ta.run_step(batches[0]).items()})
    lb.append({n: float(v) for n, v in tb.run_step_planned(batches[0]).items()})
    assert tb.plan_eager_steps == 1 and la == lb
    torch.cuda.synchronize()
    for (na, pa), (nb, pb) in zip(ta.model.named_parameters(), tb.model.named_parameters()):
        assert torch.equal(pa, pb), na


def _mapper_like(batch, k):
    """what the reference's dataset mapper hands over (dataset_mapper.py:129-259): the camera matrix as a HOST tensor whose values change
    with every sample's augmentation, plus per-sample strings / ids the step never reads"""
    out = []
    for j, d in enumerate(batch):
        d = dict(d)
        cam = d["camera_matrix"].detach().cpu().clone()
        cam[0, 0] *= 1.0 + 0.05 * k + 0.01 * j     # a resize-augmented focal length
        cam[0, 2] += 3.0 * k
        d["camera_matrix"] = cam
        d["file_name"], d["image_id"] = f"frame_{k}_{j}.png", 100 * k + j
        out.append(d)
    return out


def test_run_step_planned_follows_host_camera_matrices_that_change_per_batch():
    """a per-batch HOST tensor must not be frozen into the recording.  The mapper's camera_matrix arrives on the CPU and
    differs per batch; file_name / image_id differ per sample and must not stop the replay.  Same trajectory as run_step, bit for bit."""
    from mgnet_amd.data import synthetic_batch
    ta, _, _ = _trainer()
    tb, _, _ = _trainer()
    dev = torch.device("cuda:0")
    batches = [_mapper_like(synthetic_batch(2, 128, 256, dev, seed=60 + k), k) for k in range(7)]
    la = [{n: float(v) for n, v in ta.run_step(b).items()} for b in batches]
    lb = [{n: float(v) for n, v in tb.run_step_planned(b).items()} for b in batches]
    assert tb._plan is not None and getattr(tb, "plan_note", None) is None and getattr(tb, "plan_eager_steps", 0) == 0
    assert la == lb
    assert len({a["loss_photometric"] for a in la[4:]}) == 3, "the batches were meant to differ in their photometric loss"
    for (na, pa), (nb, pb) in zip(ta.model.named_parameters(), tb.model.named_parameters()):
        assert torch.equal(pa, pb), na
    # recording such a batch directly (no device copies) is refused before anything is frozen
    from mgnet_amd.engine.plan import PlanUnsupported
    tc, _, _ = _trainer()
    for b in batches[:3]:
        tc.run_step(b)
    with pytest.raises(PlanUnsupported, match="per-batch host tensor"):
        tc.record_plan(batches[3])


def test_a_step_the_recorder_refuses_after_it_ran_is_counted_once():
    """a `.item()` inside the step is noticed by the recorder while the body keeps running -- the step has been trained
    when the refusal surfaces; run_step_planned must not train the batch a second time (iteration, LR schedule and Adam's step count stay
    in line with an eager run)"""
    ta, batch_a, _ = _trainer()
    tb, batch_b, _ = _trainer()
    fwd = tb.model.forward

    def forward_with_a_host_read(batched_inputs):
        losses = fwd(batched_inputs)
        losses["loss_sem_seg"].item()
        return losses
    tb.model.forward = forward_with_a_host_read
    la = [{n: float(v) for n, v in ta.run_step(batch_a).items()} for _ in range(6)]
    lb = [{n: float(v) for n, v in tb.run_step_planned(batch_b).items()} for _ in range(6)]
    assert tb._plan is None and "device -> host read" in tb.plan_note
    assert ta.iter == tb.iter == 6
    assert la == lb
    for (na, pa), (nb, pb) in zip(ta.model.named_parameters(), tb.model.named_parameters()):
        assert torch.equal(pa, pb), na


def test_two_plans_of_one_step_stay_bit_identical_over_many_replays():
    """Determinism watch (round 5): two trainers with identical seeds each record their own plan and replay it side by side; every step's
    losses and flat gradient buckets must agree bit for bit.  With every struct argument's read-only declaration honoured
    (MGN_PLAN_RO=all) 1.5 - 8 % of the steps differed by one stale tile's worth -- profiles/r05_plan_determinism.txt; the default (the
    convolution family's declarations only) and MGN_PLAN_RO=none showed 0 differing steps in 1200."""
    ta, ba, _ = _trainer(seed=1)
    tb, bb, _ = _trainer(seed=1)
    for _ in range(3):
        ta.run_step(ba)
        tb.run_step(bb)
    ta.record_plan(ba)
    tb.record_plan(bb)
    for k in range(150):
        la = {n: float(v) for n, v in ta.replay_plan().items()}
        lb = {n: float(v) for n, v in tb.replay_plan().items()}
        torch.cuda.synchronize()
        assert la == lb, (k, la, lb)
        for x, y in zip(ta.reducer.buckets, tb.reducer.buckets):
            assert torch.equal(x["flat_g"], y["flat_g"]), k


def test_replay_does_not_read_the_recorded_steps_temporaries():
    """Every temporary of the step is written before it is read in EVERY replay: all blocks of the plan's private pool that are free after
    the recording (the recorded step's temporaries) are overwritten with NaN / large bit patterns before the replays -- results unchanged."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    ta, ba, _ = _trainer(seed=1)
    tb, bb, _ = _trainer(seed=1)
    for _ in range(3):
        ta.run_step(ba)
        tb.run_step(bb)
    ta.run_step(ba)
    plan = tb.record_plan(bb)
    torch.cuda.synchronize()
    pid = tuple(plan.pool.id)
    for byte in (0xFF, 0x71):
        n = 0
        for seg in torch.cuda.memory_snapshot():
            if tuple(seg.get("segment_pool_id", ())) != pid:
                continue
            a = seg["address"]
            for b in seg["blocks"]:
                if b["state"] == "inactive":
                    assert hip.hipMemset(ctypes.c_void_p(a), byte, b["size"]) == 0
                    n += 1
                a += b["size"]
        torch.cuda.synchronize()
        assert n > 20, "the plan's pool has no free blocks to poison?"
        la = {k: float(v) for k, v in ta.run_step(ba).items()}
        lb = {k: float(v) for k, v in tb.replay_plan().items()}
        assert la == lb, (hex(byte), la, lb)
    for (na, pa), (nb, pb) in zip(ta.model.named_parameters(), tb.model.named_parameters()):
        assert torch.equal(pa, pb), na


def test_read_only_declarations_are_honoured_per_family_not_per_kernel_name():
    """which struct arguments' read-only declarations a replay honours is fixed in csrc/ next to the struct (MGN_PLAN_RO_CONV =
    family 1, MGN_PLAN_RO = family 2; mgn_plan_node_ro_family), not matched on kernel names at run time.  Default mode `conv`: family 1
    only.  Checked on a recorded step: every launch whose argument struct is in family 1 is a convolution kernel, the declared structs of
    the loss / norm / input kernels are family 2, and an undeclared struct has no read-only word at all."""
    import ctypes

    from mgnet_amd import _C
    from mgnet_amd.engine import plan as plan_mod
    assert plan_mod.ro_mode() == "conv"
    t, b, _ = _trainer(seed=1)
    for _ in range(3):
        t.run_step(b)
    plan = t.record_plan(b)
    lib = _C.lib()
    fam, ro = (ctypes.c_int * 64)(), (ctypes.c_ulonglong * 128)()
    info = _C.PlanNodeInfo()
    by_family = {0: set(), 1: set(), 2: set()}
    for i in range(lib.mgn_plan_node_count(plan.handle)):
        _C.check(lib.mgn_plan_node_info(plan.handle, i, ctypes.byref(info)), "info")
        if info.type != 0:
            continue
        n = lib.mgn_plan_node_ro_family(plan.handle, i, 64, fam)
        assert n >= 0 and lib.mgn_plan_node_ro(plan.handle, i, 64, ro) == n
        for k in range(n):
            assert fam[k] in (0, 1, 2)
            assert (fam[k] != 0) == bool(ro[2 * k] | ro[2 * k + 1]), (info.name, k)
            if fam[k]:
                by_family[fam[k]].add(info.name.decode())
    conv_like = ("conv", "wgrad", "up2", "stem")
    assert by_family[1] and all(any(x in nm for x in conv_like) for nm in by_family[1]), sorted(by_family[1])
    assert any("ins_fwd" in nm for nm in by_family[2]) and any("reproj_march" in nm for nm in by_family[2]), sorted(by_family[2])
    assert not (by_family[1] & by_family[2])
    with pytest.raises(ValueError):
        plan_mod.set_ro_mode("some")


def test_two_recordings_are_checked_against_each_other_and_a_difference_drops_the_declarations(monkeypatch):
    """Trainer.record_plan(best_of, verify_steps): the two fastest recordings replay `verify_steps` steps each from one state snapshot and
    must agree bit for bit; the check leaves the trainer on the same trajectory as eager steps.  A difference (forced here) switches the
    read-only declarations off for the process, records again and checks again."""
    from mgnet_amd.engine import plan as plan_mod
    ta, ba, _ = _trainer(seed=1)
    tb, bb, _ = _trainer(seed=1)
    for _ in range(3):
        ta.run_step(ba)
        tb.run_step(bb)
    tb.record_plan(bb, best_of=2, trial_steps=2, verify_steps=5)
    chk = tb.plan_check
    assert chk["identical"] and chk["steps"] == 5 and chk["read_only_declarations"] == "conv" and chk["fallback"] is None, chk
    # steps taken by B so far: 2 recordings x (1 recorded + 2 + 2 replays), then the check's 5 steps counted ONCE (both runs start
    # from the same snapshot)
    n = 2 * 5 + 5
    assert tb.iter == 3 + n
    for _ in range(n):
        ta.run_step(ba)
    la = {k: float(v) for k, v in ta.run_step(ba).items()}
    lb = {k: float(v) for k, v in tb.replay_plan().items()}
    assert la == lb
    for (na, pa), (nb, pb) in zip(ta.model.named_parameters(), tb.model.named_parameters()):
        assert torch.equal(pa, pb), na
    # forced difference -> fallback
    real = type(tb)._verify_plans
    calls = []

    def first_one_fails(self, a, b, steps):
        out = real(self, a, b, steps)
        calls.append(plan_mod.ro_mode())
        if len(calls) == 1:
            out = dict(out, identical=False, first_difference={"step": 0, "losses": {"forced": (0.0, 1.0)}})
        return out
    monkeypatch.setattr(type(tb), "_verify_plans", first_one_fails)
    try:
        tb._plan.close()
        tb.record_plan(bb, best_of=2, trial_steps=1, verify_steps=2)
        assert calls == ["conv", "none"] and plan_mod.ro_mode() == "none"
        assert tb.plan_check["identical"] and tb.plan_check["read_only_declarations"] == "none" and tb.plan_check["fallback"]["identical"] is False
        tb.replay_plan()
    finally:
        plan_mod.set_ro_mode(None)


def test_state_snapshot_restore_repeats_a_step_bit_for_bit():
    """Trainer.state_snapshot / state_restore (what the recording check stands on): the same eager step run twice from one snapshot gives
    the same losses, gradients and parameters, fp16 loss-scale state included"""
    t, b, _ = _trainer(seed=2, dtype="float16")
    for _ in range(3):
        t.run_step(b)
    snap = t.state_snapshot()
    outs = []
    for _ in range(2):
        t.state_restore(snap)
        ls = [{k: float(v) for k, v in t.run_step(b).items()} for _ in range(3)]
        outs.append((ls, [x["flat_g"].clone() for x in t.reducer.buckets], [p.detach().clone() for p in t.model.parameters()],
                     t.optimizer.scaler.clone(), t.iter, [g["lr"] for g in t.optimizer.param_groups]))
    assert outs[0][0] == outs[1][0] and outs[0][4] == outs[1][4] and outs[0][5] == outs[1][5]
    assert all(torch.equal(x, y) for x, y in zip(outs[0][1], outs[1][1])) and all(torch.equal(x, y) for x, y in zip(outs[0][2], outs[1][2]))
    assert torch.equal(outs[0][3], outs[1][3])


def test_run_step_planned_with_the_recording_check_trains_exactly_the_eager_trajectory():
    """run_step_planned(verify_steps=k): the recording is checked (two recordings from the current state, k replays each, bit for bit)
    before it is used, and the check trains nothing -- losses of every call and the parameters afterwards equal plain eager steps"""
    ta, batch, other = _trainer(seed=3)
    tb, batch_b, other_b = _trainer(seed=3)
    la, lb = [], []
    for k in range(8):
        x, y = (batch, batch_b) if k % 2 == 0 else (other, other_b)
        la.append({n: float(v) for n, v in ta.run_step(x).items()})
        lb.append({n: float(v) for n, v in tb.run_step_planned(y, warmup=3, verify_steps=4).items()})
    assert tb._plan is not None and tb.plan_check["identical"] and tb.plan_check["steps"] == 4, (getattr(tb, "plan_note", None), tb.plan_check)
    assert ta.iter == tb.iter == 8
    assert la == lb
    for (na, pa), (nb, pb) in zip(ta.model.named_parameters(), tb.model.named_parameters()):
        assert torch.equal(pa, pb), na
