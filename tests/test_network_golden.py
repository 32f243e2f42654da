"""Network modules (group N) against golden vectors produced by the REFERENCE's own module code
(tests/golden/make_golden_network.py: mgnet/modeling/layers.py and res_net.py run unmodified on stand-ins for the absent
detectron2 / inplace_abn / fvcore names).  Pinned here:
  * the oracle's functional restatement (oracle/network_oracle.py)            -- CPU
  * the product's host mirror: same constructor arguments, the SAME state-dict keys (strict load), same outputs and
    running statistics                                                          -- CPU (host logic) and GPU (HIP kernels, bf16)
"""
import importlib.util
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_golden_network", os.path.join(HERE, "golden", "make_golden_network.py"))
G = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(G)


def _product(case):
    from mgnet_amd.modeling import layers, res_net
    from mgnet_amd.registry import ShapeSpec
    ns = type("NS", (), {})()
    for m in (layers, res_net):
        for k in dir(m):
            setattr(ns, k, getattr(m, k))
    return G.build(ns, ShapeSpec, case)


def _golden(case):
    return np.load(os.path.join(HERE, "golden", f"net_{case}.npz"))


def _run(m, x):
    y = m(x) if isinstance(x, dict) else m(*x)
    return G.flatten_out(y)


def _close(a, ref, tol):
    ref = torch.from_numpy(ref)
    return float((a.detach().float().cpu() - ref).abs().max() / (ref.abs().max() + 1e-12)) < tol


@pytest.mark.parametrize("case", sorted(G.CASES))
def test_product_host_mirror_matches_reference_modules_cpu(case):
    g = _golden(case)
    m = _product(case)
    assert sorted(m.state_dict().keys()) == list(g["keys"]), "state-dict keys differ from the reference module's"
    G.fill_state(m, seed=1)
    m.train()
    x = G.make_inputs(case)
    outs = _run(m, x)
    assert len(outs) == len([k for k in g.files if k.startswith("out")])
    for i, o in enumerate(outs):
        assert o.shape == g[f"out{i}"].shape and _close(o, g[f"out{i}"], 2e-4), (case, i)
    sd = m.state_dict()
    for i, k in enumerate(g["run_keys"]):   # momentum 0.01, unbiased variance
        assert torch.allclose(sd[str(k)], torch.from_numpy(g[f"run{i}"]), rtol=1e-4, atol=1e-5), (case, str(k))


@pytest.mark.parametrize("case", sorted(G.CASES))
def test_network_oracle_matches_reference_modules(case):
    from oracle import network_oracle as O

    g = _golden(case)
    sd = {"m." + k: v for k, v in G.fill_state(_product(case), seed=1).items()}
    x = G.make_inputs(case)
    c = G.CASES[case]
    if c["kind"] == "BasicBlock":
        outs = [O.basic_block(sd, "m", x[0], c["kw"]["stride"])]
    elif c["kind"] == "BasicStem":
        outs = [F.max_pool2d(O.conv_abn(sd, "m.conv1", x[0], 2, 3), kernel_size=3, stride=2, padding=1)]
    elif c["kind"] == "GlobalContextModule":
        outs = [O.gcm({k.replace("m.", "global_context.", 1): v for k, v in sd.items()}, x[0])]
    elif c["kind"] == "AttentionRefinementModule":
        outs = [O.arm(sd, "m", x[0])]
    elif c["kind"] == "FeatureFusionModule":
        outs = [O.ffm(sd, "m", x[0], x[1])]
    elif c["kind"] == "MGNetHead":
        outs = [O.head(sd, "m", x[0])]
    else:
        y, msc = O.decoder(sd, "m", x)
        outs = [y] + list(msc)
    for i, o in enumerate(outs):
        assert _close(o, g[f"out{i}"], 2e-4), (case, i)


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(G.CASES))
def test_product_hip_path_matches_reference_modules(case):
    g = _golden(case)
    m = _product(case)
    G.fill_state(m, seed=1)
    m = m.cuda().train()
    x = G.make_inputs(case)

    def dev(t):
        t = t.cuda().to(torch.bfloat16)
        if G.CASES[case]["kind"] == "BasicStem":   # the stem consumes the channel-padded input of csrc/prep.hip
            t = torch.cat([t, t.new_zeros(t.shape[0], 5, *t.shape[2:])], 1)
        return t.contiguous(memory_format=torch.channels_last)
    x = {k: dev(v) for k, v in x.items()} if isinstance(x, dict) else [dev(v) for v in x]
    outs = _run(m, x)
    for i, o in enumerate(outs):
        assert o.dtype == torch.bfloat16 and _close(o, g[f"out{i}"], 4e-2), (case, i)   # bf16 activations end to end
