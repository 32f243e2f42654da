import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Load tests/golden/<name>.npz -> (inputs dict, outputs dict)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    ins = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    outs = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    return ins, outs


def golden_case_inputs(name):
    """Inputs of a reprojection golden case as a dict(inv=[..], img, prev, nxt, poses, K, mask)."""
    sys.path.insert(0, GOLDEN)
    try:
        import make_golden
    finally:
        sys.path.pop(0)
    if name in make_golden.SEED_ONLY:
        return make_golden.build_case(name)
    ins, _ = load_golden("reproj_" + name)
    c = {k: ins[k] for k in ("img", "prev", "nxt", "poses", "K")}
    c["inv"] = [ins[f"inv{i}"] for i in range(3)]
    c["mask"] = ins.get("mask")
    return c


REPROJ_CASES = ["rand_small", "smooth", "identity_pose", "oob_clamp", "no_mask_odd"]


@pytest.fixture
def torch_staging(monkeypatch):
    """fp32 activations on the GPU: the product's convolution kernels are bf16, so an fp32 network runs its convolutions on torch's
    (staging); mgnet_amd refuses that unless explicitly allowed -- tests that exercise fp32 accuracy of everything else opt in"""
    monkeypatch.setenv("MGNET_ALLOW_TORCH_STAGING", "1")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    import oracle

    oracle.build()


# ---- panoptic target fixtures (tests/golden/targets_*.npz, outputs of the reference's target_generator.py) -------------
import glob  # noqa: E402

TARGET_GOLD = sorted(glob.glob(os.path.join(GOLDEN, "targets_*.npz")))


def load_target_case(path):
    """-> (npz, label image, segments_info, generator kwargs)"""
    z = np.load(path)
    kw = {str(k): int(v) for k, v in zip(z["gen_keys"], z["gen_vals"])}
    for k in ("ignore_stuff_in_offset", "ignore_crowd_in_semantic"):
        if k in kw:
            kw[k] = bool(kw[k])
    kw["thing_ids"] = [int(t) for t in z["thing_ids"]]
    segs = [dict(id=int(r[0]), category_id=int(r[1]), iscrowd=int(r[2])) for r in z["segments"]]
    return z, z["panoptic"], segs, kw


def golden_script(name):
    """Import tests/golden/<name>.py by path (for its synthetic-input builders)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(GOLDEN, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m
