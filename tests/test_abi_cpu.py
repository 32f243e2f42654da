"""CPU: the C-ABI library loads and exports every symbol include/mgnet_hip.h declares; argument validation
(no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from mgnet_amd import _C, build

    build.build()
    L = _C.lib()
    hdr = open(os.path.join(ROOT, "include", "mgnet_hip.h")).read()
    declared = set(re.findall(r"\b(mgn_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_C.SYMBOLS) | set(_C.SYMBOLS_F16), (declared ^ (set(_C.SYMBOLS) | set(_C.SYMBOLS_F16)))
    for s in declared:
        assert hasattr(L, s), f"libmgnet_hip.so does not export {s}"
    assert L.mgn_version().decode().startswith("mgnet_hip")


def test_workspace_query_and_validation():
    from mgnet_amd import _C

    cfg = _C.make_reproj_cfg(8, 1024, 2048, 3)
    assert 0 < _C.reproj_workspace_bytes(cfg) < (64 << 20)
    for bad in (dict(B=0), dict(H=1), dict(W=1), dict(n_scales=0), dict(n_scales=5)):
        kw = dict(B=2, H=8, W=8, n_scales=3)
        kw.update(bad)
        with pytest.raises(RuntimeError):
            _C.reproj_workspace_bytes(_C.make_reproj_cfg(**kw))
    # invalid option combinations are refused before any launch
    cfg = _C.make_reproj_cfg(1, 8, 8, 3, ssim_w=-0.1)     # not a weight
    rc = _C.lib().mgn_reproj_loss_fwd(ctypes.byref(cfg), *([None] * 6), 16, 4, None, 0, *([None] * 5), 0, None)
    assert rc == -22
    cfg = _C.make_reproj_cfg(1, 8, 8, 3, automask=True, reduce_op="mean")   # the reference asserts (loss.py:105-109)
    rc = _C.lib().mgn_reproj_loss_fwd(ctypes.byref(cfg), *([None] * 6), 16, 4, None, 0, *([None] * 5), 0, None)
    assert rc == -22


def test_product_package_never_imports_the_oracle():
    import subprocess, sys
    code = "import sys; import mgnet_amd, mgnet_amd._C, mgnet_amd.modeling; assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mgnet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_integration_md_reproj_cfg_matches_the_header():
    """INTEGRATION.md section 2(b) shows the ctypes struct a maintainer would write for mgn_reproj_cfg: its field list must be the
    header's (a missing field only 'worked' while it fell into alignment padding)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "mgnet_hip.h")).read()
    body = hdr[hdr.index("typedef struct {", hdr.index("Self-supervised photometric reprojection loss")):hdr.index("} mgn_reproj_cfg;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields_h = re.findall(r"(?:int|float|void\s*\*)\s*([A-Za-z_, ]+);", body)
    fields_h = [f.strip() for grp in fields_h for f in grp.split(",")]
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    cfg = md[md.index("class Cfg(ctypes.Structure):"):md.index("def ptrs(ts):")]
    fields_md = re.findall(r'\("(\w+)",\s*ctypes', cfg)
    assert fields_md == fields_h, (fields_md, fields_h)
    from mgnet_amd import _C
    assert [f[0] for f in _C.ReprojCfg._fields_] == fields_h
