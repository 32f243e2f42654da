#!/usr/bin/env python3
"""Golden vectors for the NETWORK modules (group N) produced by running the reference's own module code.

Runs only in the build container (needs /root/reference); only the .npz fixtures travel.  The reference files
mgnet/modeling/layers.py and res_net.py are imported unmodified; the three third-party packages they import are absent
from the image and from /root/reference, so this harness provides stand-ins for exactly the names they use:

  * detectron2.layers.Conv2d        nn.Conv2d followed by the optional `norm` and `activation` submodules (detectron2 v0.6)
  * detectron2.layers.ShapeSpec / CNNBlockBase, detectron2.modeling.BACKBONE_REGISTRY / ResNet: containers only
  * fvcore.nn.weight_init           initialisers (irrelevant here: every parameter is overwritten deterministically)
  * inplace_abn.InPlaceABNSync      (pip inplace-abn >= 1.1.0) its published forward semantics in plain torch:
        y = act(batch_norm(x; gamma = |weight| + eps, beta = bias)), act = leaky_relu(0.01) | identity,
        running statistics with `momentum` and the unbiased variance.
What the fixtures therefore PIN is the reference's wiring -- which layers exist under which state-dict keys, their
hyper-parameters, the order of operations in every forward(), interpolation modes, concatenation order, residual and
attention arithmetic -- not the internals of inplace_abn, which stay a restatement.

Parameters are not stored: `fill_state(module, seed)` regenerates them from numpy RandomState streams keyed by the
parameter name, and tests/test_network_golden.py calls the same function.

Usage:  python tests/golden/make_golden_network.py        (rewrites tests/golden/net_*.npz)
"""
import os
import sys
import types
import zlib
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference/mgnet"
OUT = os.path.dirname(os.path.abspath(__file__))


def fill_state(module, seed):
    """Deterministic parameters/buffers: one RandomState per (seed, key).  Shared with the tests."""
    sd = module.state_dict()
    new = {}
    for k in sorted(sd):
        v = sd[k]
        rs = np.random.RandomState((zlib.crc32(k.encode()) + 7919 * seed) % (2 ** 31))
        if k.endswith("num_batches_tracked"):
            new[k] = v.clone()
        elif k.endswith("running_mean"):
            new[k] = torch.from_numpy(rs.normal(0, 0.1, tuple(v.shape)).astype(np.float32))
        elif k.endswith("running_var"):
            new[k] = torch.from_numpy(rs.uniform(0.5, 1.5, tuple(v.shape)).astype(np.float32))
        elif v.dim() == 1 and k.endswith("norm.weight") or (v.dim() == 1 and ".weight" in k and "norm" in k):
            w = rs.uniform(0.5, 1.5, tuple(v.shape)).astype(np.float32)
            w[::3] *= -1.0  # negative gammas: InPlaceABN uses |weight|
            new[k] = torch.from_numpy(w)
        elif v.dim() == 1:
            new[k] = torch.from_numpy(rs.normal(0, 0.1, tuple(v.shape)).astype(np.float32))
        else:
            fan_in = int(np.prod(v.shape[1:]))
            new[k] = torch.from_numpy(rs.normal(0, (2.0 / fan_in) ** 0.5, tuple(v.shape)).astype(np.float32))
    module.load_state_dict(new, strict=True)
    return new


def install_stand_ins():
    ShapeSpec = namedtuple("ShapeSpec", ["channels", "height", "width", "stride"], defaults=(None, None, None, None))

    class Conv2d(nn.Conv2d):
        def __init__(self, *a, **k):
            norm, act = k.pop("norm", None), k.pop("activation", None)
            super().__init__(*a, **k)
            self.norm, self.activation = norm, act

        def forward(self, x):
            x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
            if self.norm is not None:
                x = self.norm(x)
            if self.activation is not None:
                x = self.activation(x)
            return x

    class CNNBlockBase(nn.Module):
        def __init__(self, in_channels, out_channels, stride):
            super().__init__()
            self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride

    class InPlaceABNSync(nn.Module):
        def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, activation="leaky_relu", activation_param=0.01, group=None):
            super().__init__()
            self.eps, self.momentum, self.activation, self.activation_param = eps, momentum, activation, activation_param
            self.weight, self.bias = nn.Parameter(torch.ones(num_features)), nn.Parameter(torch.zeros(num_features))
            self.register_buffer("running_mean", torch.zeros(num_features))
            self.register_buffer("running_var", torch.ones(num_features))

        def forward(self, x):
            y = F.batch_norm(x, self.running_mean, self.running_var, self.weight.abs() + self.eps, self.bias, self.training,
                             self.momentum, self.eps)
            return F.leaky_relu(y, self.activation_param) if self.activation == "leaky_relu" else y

    class _Registry:
        def register(self, obj=None):
            return obj if obj is not None else (lambda o: o)

        def get(self, name):
            raise KeyError(name)

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("detectron2")
    mod("detectron2.layers", Conv2d=Conv2d, ShapeSpec=ShapeSpec, CNNBlockBase=CNNBlockBase)
    mod("detectron2.modeling", BACKBONE_REGISTRY=_Registry(), ResNet=type("ResNet", (nn.Module,), {}))
    mod("inplace_abn", InPlaceABNSync=InPlaceABNSync)
    mod("fvcore")
    mod("fvcore.nn")
    wi = mod("fvcore.nn.weight_init", c2_msra_fill=lambda m: None, c2_xavier_fill=lambda m: None)
    sys.modules["fvcore.nn"].weight_init = wi
    return ShapeSpec


def import_reference_modules():
    ShapeSpec = install_stand_ins()
    for name, path in (("mgnet", REF), ("mgnet.modeling", REF + "/modeling")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    import mgnet.modeling.layers as RL  # noqa
    import mgnet.modeling.res_net as RR  # noqa
    return RL, RR, ShapeSpec


def randn(seed, *shape):
    return torch.from_numpy(np.random.RandomState(seed).normal(0, 1, shape).astype(np.float32))


# name -> (constructor kwargs understood by BOTH the reference and mgnet_amd, input shapes)
CASES = {
    "block_s1": dict(kind="BasicBlock", args=(32, 32), kw=dict(stride=1), inputs=[(2, 32, 10, 12)]),
    "block_s2": dict(kind="BasicBlock", args=(32, 64), kw=dict(stride=2), inputs=[(2, 32, 11, 14)]),
    "stem": dict(kind="BasicStem", args=(3, 32), kw={}, inputs=[(2, 3, 30, 36)]),
    "gcm": dict(kind="GlobalContextModule", args=(64, 32), kw={}, inputs=[(3, 64, 4, 6)]),
    "arm": dict(kind="AttentionRefinementModule", args=(64, 32), kw={}, inputs=[(2, 64, 6, 8)]),
    "ffm": dict(kind="FeatureFusionModule", args=(64, 32), kw={}, inputs=[(2, 32, 8, 10), (2, 32, 8, 10)]),
    "head": dict(kind="MGNetHead", args=(32, 32, 19), kw={}, inputs=[(2, 32, 8, 10)]),
    "decoder": dict(kind="MGNetDecoder", args=(), kw=dict(common_stride=8, arm_channels=[32, 32], refine_channels=[32, 32],
                                                          ffm_channels=64),
                    shapes={"res3": (32, 8), "res4": (64, 16), "res5": (96, 32)},
                    inputs={"res3": (2, 32, 12, 16), "res4": (2, 64, 6, 8), "res5": (2, 96, 3, 4), "global_context": (2, 32, 3, 4)}),
}


def build(ns, ShapeSpec, case):
    c = CASES[case]
    cls = getattr(ns, c["kind"])
    if c["kind"] == "MGNetDecoder":
        shape = {k: ShapeSpec(channels=ch, stride=st) for k, (ch, st) in c["shapes"].items()}
        return cls(shape, **c["kw"])
    return cls(*c["args"], **c["kw"])


def make_inputs(case):
    c = CASES[case]
    base = zlib.crc32(case.encode()) % 100000
    if isinstance(c["inputs"], dict):
        return {k: randn(base + i, *s) for i, (k, s) in enumerate(sorted(c["inputs"].items()))}
    return [randn(base + i, *s) for i, s in enumerate(c["inputs"])]


def flatten_out(y):
    if isinstance(y, torch.Tensor):
        return [y]
    out = []
    for v in y:
        out += flatten_out(v)
    return out


def main():
    RL, RR, ShapeSpec = import_reference_modules()
    for case, c in CASES.items():
        ns = RR if c["kind"] in ("BasicBlock", "BasicStem") else RL
        m = build(ns, ShapeSpec, case).double()
        fill_state(m, seed=1)
        m = m.double().train()
        x = make_inputs(case)
        with torch.no_grad():
            if isinstance(x, dict):
                y = m({k: v.double() for k, v in x.items()})
            else:
                y = m(*[v.double() for v in x])
        outs = flatten_out(y)
        sd = m.state_dict()
        arrays = {f"out{i}": o.float().numpy() for i, o in enumerate(outs)}
        arrays["keys"] = np.array(sorted(sd.keys()))
        # running statistics after one training forward (momentum 0.01, unbiased variance)
        rk = sorted(k for k in sd if k.endswith("running_mean") or k.endswith("running_var"))
        for i, k in enumerate(rk):
            arrays[f"run{i}"] = sd[k].float().numpy()
        arrays["run_keys"] = np.array(rk)
        np.savez_compressed(os.path.join(OUT, f"net_{case}.npz"), **arrays)
        print(case, [tuple(o.shape) for o in outs], len(sd), "state keys")


if __name__ == "__main__":
    main()
