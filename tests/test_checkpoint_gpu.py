"""GPU: `.pth` save / resume with the fused optimizer (flat parameter + moment buffers) and its torch.optim.Adam-compatible
state layout."""
import pytest
import torch
from test_network_cpu import small_model

pytestmark = pytest.mark.gpu


def test_resume_reproduces_the_trajectory(tmp_path, monkeypatch):
    # the bf16 HIP step is bit-reproducible (no order-dependent arithmetic), so the resumed trajectory has to be
    # THE trajectory (the fp32 torch-staging convolutions this test used before are not reproducible run to run, and a 2e-3
    # tolerance failed about once in ten runs)
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer
    from mgnet_amd.solver.fused_adam import FusedAdam

    cfg, m = small_model(with_depth=True, seed=1)
    cfg.OUTPUT_DIR = str(tmp_path)
    m.amp_dtype = torch.bfloat16
    tr = Trainer(cfg, m.cuda())
    assert isinstance(tr.optimizer, FusedAdam)
    batch = synthetic_batch(2, 64, 96, "cuda", seed=2)
    for _ in range(3):
        tr.run_step(batch)
    path = tr.save()
    blob = torch.load(path, map_location="cpu", weights_only=False)
    assert blob["iteration"] == 2 and set(blob) == {"model", "optimizer", "scheduler", "iteration"}
    # the optimizer entry loads into a plain torch.optim.Adam over the same groups (what the reference would do)
    groups = [{"params": [torch.nn.Parameter(torch.zeros_like(p)) for p in g["params"]], "lr": g["lr"]} for g in tr.optimizer.param_groups]
    adam = torch.optim.Adam(groups, lr=1e-4)
    adam.load_state_dict(blob["optimizer"])
    p0 = tr.optimizer.param_groups[0]["params"][0]
    st = adam.state[adam.param_groups[0]["params"][0]]
    assert float(st["step"]) == 3 and st["exp_avg"].shape == p0.shape and float(st["exp_avg_sq"].abs().sum()) > 0
    ref = [float(sum(v.detach() for v in tr.run_step(batch).values())) for _ in range(3)]

    cfg2, m2 = small_model(with_depth=True, seed=9)
    cfg2.OUTPUT_DIR = str(tmp_path)
    m2.amp_dtype = torch.bfloat16
    tr2 = Trainer(cfg2, m2.cuda())
    tr2.resume_or_load(resume=True)
    assert tr2.iter == 3 and tr2.optimizer._t == 3
    for k, v in tr2.model.state_dict().items():
        assert torch.equal(v.cpu(), blob["model"][k]), k
    got = [float(sum(v.detach() for v in tr2.run_step(batch).values())) for _ in range(3)]
    assert got == pytest.approx(ref, rel=1e-6), (got, ref)


def test_fp16_resume_restores_the_loss_scaler(tmp_path):
    """fp16 + dynamic loss scaling: the scale, its growth tracker and the DEVICE-side count of the steps actually taken (what Adam's
    bias corrections continue from) travel with the checkpoint (detectron2's AMPTrainer checkpoints `grad_scaler`); without them a
    resumed run restarts the bias corrections at t = 1 against trained moments."""
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer

    cfg, m = small_model(with_depth=True, seed=1)
    cfg.OUTPUT_DIR = str(tmp_path)
    m.amp_dtype = torch.float16
    tr = Trainer(cfg, m.cuda())
    assert tr.optimizer.scaler is not None
    batch = synthetic_batch(2, 64, 96, "cuda", seed=2)
    for _ in range(5):
        tr.run_step(batch)
    sc = [float(v) for v in tr.optimizer.scaler.tolist()]
    path = tr.save()
    blob = torch.load(path, map_location="cpu", weights_only=False)
    gs = blob["optimizer"]["grad_scaler"]
    assert gs["scale"] == sc[0] and gs["growth_tracker"] == int(sc[1]) and gs["steps_taken"] == int(sc[2]) and gs["host_steps"] == 5
    assert all(float(st["step"]) == sc[2] for st in blob["optimizer"]["state"].values())   # steps TAKEN, not steps attempted
    ref = [float(sum(v.detach() for v in tr.run_step(batch).values())) for _ in range(3)]
    ref_sc = [float(v) for v in tr.optimizer.scaler.tolist()]

    cfg2, m2 = small_model(with_depth=True, seed=9)
    cfg2.OUTPUT_DIR = str(tmp_path)
    m2.amp_dtype = torch.float16
    tr2 = Trainer(cfg2, m2.cuda())
    tr2.resume_or_load(resume=True)
    assert [float(v) for v in tr2.optimizer.scaler.tolist()] == sc and tr2.optimizer._t == 5
    got = [float(sum(v.detach() for v in tr2.run_step(batch).values())) for _ in range(3)]
    assert got == pytest.approx(ref, rel=1e-6), (got, ref)
    assert [float(v) for v in tr2.optimizer.scaler.tolist()] == ref_sc

    # a torch.optim.Adam state without a scaler entry (a reference checkpoint): the bias corrections continue from its step count
    del blob["optimizer"]["grad_scaler"]
    tr2.optimizer.load_state_dict(blob["optimizer"])
    assert float(tr2.optimizer.scaler[2]) == sc[2]
