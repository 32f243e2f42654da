"""Checkpoint import/export (mgnet_amd/checkpoint.py): key renaming against the reference's own convert_key
(tests/golden/checkpoint_keys.json), ImageNet `.pkl` import with the suffix heuristic, `.pth` round trip."""
import json
import os
import pickle

import numpy as np
import pytest
import torch
from conftest import GOLDEN, golden_script
from test_network_cpu import small_model

from mgnet_amd.checkpoint import Checkpointer, convert_key, convert_torchvision_resnets

MK = golden_script("make_golden_checkpoint")


def test_convert_key_matches_reference():
    gold = json.load(open(os.path.join(GOLDEN, "checkpoint_keys.json")))
    assert set(gold) == {"resnet18/backbone", "resnet18/pose_encoder", "resnet34/backbone", "resnet34/pose_encoder"}
    for name, table in gold.items():
        prefix = name.split("/")[1]
        assert len(table) >= 122
        for k, want in table.items():
            assert convert_key(k, prefix) == want, k


def fake_torchvision_resnet18(seed):
    """A state dict with torchvision's ResNet-18 names and shapes, random values."""
    g = torch.Generator().manual_seed(seed)
    ch = {0: 64, 1: 64, 2: 128, 3: 256, 4: 512}
    sd = {}
    for k in MK.torchvision_resnet_keys():
        p = k.split(".")
        if k == "conv1.weight":
            shape = (64, 3, 7, 7)
        elif p[0] == "bn1":
            shape = (64,)
        elif p[0] == "fc":
            shape = (1000, 512) if p[1] == "weight" else (1000,)
        else:
            li, b = int(p[0][5:]), int(p[1])
            cout = ch[li]
            cin = ch[li - 1] if b == 0 else cout
            if p[2] == "conv1":
                shape = (cout, cin, 3, 3)
            elif p[2] == "conv2":
                shape = (cout, cout, 3, 3)
            elif p[2] == "downsample" and p[3] == "0":
                shape = (cout, cin, 1, 1)
            else:
                shape = (cout,)
        sd[k] = torch.tensor(7) if k.endswith("num_batches_tracked") else torch.randn(shape, generator=g)
    return sd


def test_imagenet_pkl_import(tmp_path):
    tv_a, tv_b = fake_torchvision_resnet18(1), fake_torchvision_resnet18(2)
    blob = convert_torchvision_resnets(tv_a, tv_b)
    assert blob["__author__"] == "torchvision" and blob["matching_heuristics"] is True
    assert all(isinstance(v, np.ndarray) for v in blob["model"].values())
    path = str(tmp_path / "imagenet_weights.pkl")
    with open(path, "wb") as f:
        pickle.dump(blob, f)
    cfg, m = small_model()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    ck = Checkpointer(m)
    assert ck.load(path) == {}
    sd = m.state_dict()
    assert torch.equal(sd["backbone.stem.conv1.weight"], tv_a["conv1.weight"])
    assert torch.equal(sd["backbone.stem.conv1.norm.running_var"], tv_a["bn1.running_var"])
    assert torch.equal(sd["backbone.res3.0.shortcut.weight"], tv_a["layer2.0.downsample.0.weight"])
    assert torch.equal(sd["backbone.res5.1.conv2.norm.bias"], tv_a["layer4.1.bn2.bias"])
    # the pose encoder: found through the suffix heuristic, 9-channel stem = the 3-channel weight x3 / 3
    assert torch.equal(sd["pose_net.pose_encoder.res4.1.conv1.weight"], tv_b["layer3.1.conv1.weight"])
    assert torch.equal(sd["pose_net.pose_encoder.stem.conv1.weight"], torch.cat([tv_b["conv1.weight"]] * 3, 1) / 3)
    inc = ck.last_incompatible
    assert "backbone.stem.fc.weight" in inc.unexpected_keys and "backbone.stem.conv1.norm.num_batches_tracked" in inc.unexpected_keys
    assert not inc.shape_mismatch
    assert "sem_seg_head.head.predictor.weight" in inc.missing_keys and not any(k.startswith("backbone.") for k in inc.missing_keys)
    assert torch.equal(sd["sem_seg_head.head.predictor.weight"], before["sem_seg_head.head.predictor.weight"])


def test_pth_round_trip_and_resume(tmp_path):
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer

    cfg, m = small_model(with_depth=False, seed=1)
    cfg.OUTPUT_DIR = str(tmp_path)
    tr = Trainer(cfg, m)
    batch = synthetic_batch(1, 64, 96, "cpu", seed=2, with_depth=False)
    for _ in range(2):
        tr.run_step(batch)
    path = tr.save()
    blob = torch.load(path, map_location="cpu", weights_only=False)
    assert set(blob) == {"model", "optimizer", "scheduler", "iteration"} and blob["iteration"] == 1   # DetectionCheckpointer layout
    assert list(blob["model"].keys()) == list(m.state_dict().keys())
    # continue two more steps -> reference trajectory
    ref = [float(sum(tr.run_step(batch).values())) for _ in range(2)]
    # a fresh trainer resumes from the file and reproduces it
    cfg2, m2 = small_model(with_depth=False, seed=5)
    cfg2.OUTPUT_DIR = str(tmp_path)
    tr2 = Trainer(cfg2, m2)
    tr2.resume_or_load(resume=True)
    assert tr2.iter == 2 and tr2.scheduler.last_epoch == tr.scheduler.last_epoch - 2
    got = [float(sum(tr2.run_step(batch).values())) for _ in range(2)]
    assert got == pytest.approx(ref, rel=1e-5)
    # resume=False loads MODEL.WEIGHTS only (empty here) and keeps the iteration at 0
    tr3 = Trainer(cfg2, small_model(with_depth=False, seed=6)[1])
    assert tr3.resume_or_load(resume=False) == {} and tr3.iter == 0


def test_shape_mismatch_is_reported_not_loaded(tmp_path):
    cfg, m = small_model(with_depth=False)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["sem_seg_head.head.predictor.weight"] = torch.zeros(5, 256, 1, 1)
    path = str(tmp_path / "x.pth")
    torch.save({"model": sd}, path)
    ck = Checkpointer(m)
    ck.load(path)
    assert [x[0] for x in ck.last_incompatible.shape_mismatch] == ["sem_seg_head.head.predictor.weight"]
    assert ck.last_incompatible.missing_keys == ["sem_seg_head.head.predictor.weight"]
