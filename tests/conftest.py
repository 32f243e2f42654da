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


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    import oracle

    oracle.build()
