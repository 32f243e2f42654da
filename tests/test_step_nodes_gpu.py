"""GPU: every convolution / InPlaceABN node of a REAL full training step checked against fp64, with the fused pieces in the
configuration the composed step uses them in.

tests/test_conv_gpu.py and test_iabn_gpu.py check each kernel on its own; the full-step gradient comparison of
tests/test_grad_parity_gpu.py is bounded from below by what bf16 costs this network (about 60 norm layers whose backward
cancellations amplify every 2^-9 rounding: a plain-torch bf16 evaluation is 0.4 away in relative error as well).  What was never
checked sharply is the COMPOSITION: the ResNet shortcut gradient added inside the data-gradient kernel (`with_skip`), the
channel-padded predictors (`cout_pad`), the batch statistics taken from the producing convolution's epilogue (`stats_for`), the
norm backward that re-derives x_hat from the stored output, the block-tail / stem fusions.  Here one real step of the whole model
runs on the bf16 HIP path while every `_ConvFn`, `_IABNFn`, `_AbnAddReluFn` and `_AbnPoolFn` node records its actual inputs,
outputs and gradients; each node is then re-evaluated in fp64 FROM ITS OWN RECORDED INPUTS (mg_net.py:249-373 wiring,
res_net.py:68-79, layers.py:22-127 as they occur).  A node's error is one bf16 rounding of its result plus fp32 accumulation --
nothing is amplified -- so the bounds are sharp: a wrong fused term is an O(1) error in that node."""
import pytest
import torch
import torch.nn.functional as F

from test_network_cpu import small_model
from test_network_gpu import _randomise

pytestmark = pytest.mark.gpu


def _snap(v):
    return v.detach().clone() if torch.is_tensor(v) else v


def _record(monkeypatch, cls, log):
    f0, b0 = cls.forward, cls.backward

    def fwd(ctx, *a):
        ins = [_snap(v) for v in a]          # before the call: the in-place norm overwrites its input
        out = f0(ctx, *a)
        ctx._node = dict(kind=cls.__name__, ins=ins, outs=[_snap(v) for v in (out if isinstance(out, tuple) else (out,))])
        return out

    def bwd(ctx, *g):
        gout = [_snap(v) for v in g]         # before the call: backward kernels may consume their gradient in place
        gi = b0(ctx, *g)
        log.append(dict(ctx._node, gout=gout, gin=[_snap(v) for v in (gi if isinstance(gi, tuple) else (gi,))],
                        needs=tuple(ctx.needs_input_grad)))
        return gi

    monkeypatch.setattr(cls, "forward", staticmethod(fwd))
    monkeypatch.setattr(cls, "backward", staticmethod(bwd))


def _rel(got, ref):
    ref = ref.double()
    return float((got.double() - ref).norm() / (ref.norm() + 1e-30))


def _q(t, dtype):
    return t.to(dtype).double()


def _check_conv(n, dtype, tol_h, tol_f):
    x, w, bias, stride, pad, relu, with_skip, cout_pad = n["ins"][:8]
    cin = w.shape[1]
    xd = x[:, :cin].double().requires_grad_(True)                 # (channel-padded stem input: the real channels)
    wd = _q(w, dtype).requires_grad_(True)
    bd = None if bias is None else bias.double().requires_grad_(True)
    y = F.conv2d(xd, wd, bd, stride=stride, padding=pad)
    if relu:
        y = torch.relu(y)
    cout = w.shape[0]
    out = n["outs"][0]
    if out.shape[1] != cout:     # keep_pad: the node works on the 32-channel-padded map; padding channels are zero in both directions
        assert float(out[:, cout:].abs().max()) == 0.0 and float(n["gout"][0][:, cout:].abs().max()) == 0.0
        out = out[:, :cout]
    errs = {"y": _rel(out, y)}
    assert errs["y"] < tol_h, ("conv forward", tuple(w.shape), stride, errs)
    dy = n["gout"][0][:, :cout]
    if relu:   # the recorded output is the rounded activation: its zero pattern is the mask the kernel used
        y = y * 0 + F.conv2d(xd, wd, bd, stride=stride, padding=pad) * (out > 0)
    y.backward(dy.double())
    dx, dw, db = n["gin"][:3]
    if dx is not None:
        ref = xd.grad.clone()
        if len(n["gout"]) > 1 and n["gout"][1] is not None:   # fused shortcut gradient
            gs = n["gout"][1].double()
            if gs.shape == ref.shape:
                ref += gs
            else:            # with_skip = 2: the 1x1 / stride-2 shortcut's gradient at the LOW resolution, belonging to the even pixels
                assert with_skip == 2 and gs.shape == ref[:, :, ::2, ::2].shape
                ref[:, :, ::2, ::2] += gs
        errs["dx"] = _rel(dx, ref)
        assert errs["dx"] < tol_h, ("conv data gradient", tuple(w.shape), stride, "skip" if with_skip else "", errs)
    if dw is not None:
        errs["dw"] = _rel(dw, wd.grad)
        assert errs["dw"] < tol_f, ("conv weight gradient", tuple(w.shape), stride, errs)
    if db is not None:
        errs["db"] = _rel(db, bd.grad)
        assert errs["db"] < tol_f, ("conv bias gradient", tuple(w.shape), errs)
    return errs


def _check_shortcut_s2(n, dtype, tol_h, tol_f):
    """ops._ShortcutS2Fn: the 1x1 / stride-2 shortcut conv fed with conv1's `xsub`: forward = the strided conv of the full input, the
    data gradient = that conv's input gradient AT the pixels it reads (the rest is zero and never formed)"""
    xsub, xfull, w = n["ins"][:3]
    xd = xfull.double().requires_grad_(True)
    wd = _q(w, dtype).requires_grad_(True)
    y = F.conv2d(xd, wd, None, stride=2, padding=0)
    errs = {"y": _rel(n["outs"][0], y)}
    assert errs["y"] < tol_h, ("shortcut conv forward", tuple(w.shape), errs)
    y.backward(n["gout"][0].double())
    dsub, _, dw = n["gin"][:3]
    assert tuple(dsub.shape) == tuple(xsub.shape)
    rest = xd.grad.clone()
    rest[:, :, ::2, ::2] = 0
    assert float(rest.abs().max()) == 0.0
    errs["dx"] = _rel(dsub, xd.grad[:, :, ::2, ::2])
    assert errs["dx"] < tol_h, ("shortcut conv data gradient", tuple(w.shape), errs)
    errs["dw"] = _rel(dw, wd.grad)
    assert errs["dw"] < tol_f, ("shortcut conv weight gradient", tuple(w.shape), errs)
    return errs


def _bn_act(xd, wd, bd, eps, activation, slope):
    y = F.batch_norm(xd, None, None, wd.abs() + eps, bd, True, 0.0, eps)
    return F.leaky_relu(y, slope) if activation == "leaky_relu" else y


def _check_norm(n, tol_h, tol_p, tol_dx):
    kind = n["kind"]
    if kind == "_AbnAddReluFn":
        x, shortcut, w, b = n["ins"][:4]
        eps, activation, slope = n["ins"][8], "identity", 0.01
    else:
        x, w, b = n["ins"][:3]
        shortcut, eps, activation, slope = None, n["ins"][7], n["ins"][8], n["ins"][9]
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    y = _bn_act(xd, wd, bd, eps, activation, slope)
    sd = None
    if kind == "_AbnAddReluFn":
        # the fused block tail keeps the rounding points of the separate ops (norm output stored in 16 bits, THEN + shortcut, ReLU):
        # the ReLU mask is decided on that rounded sum
        sd = shortcut.double().requires_grad_(True)
        yq = y + (y.detach().to(x.dtype).double() - y.detach())
        y = torch.relu(yq + sd)
    elif kind == "_AbnPoolFn":
        # pool the ROUNDED activations like the kernel does (arg-max ties are decided on the stored 16-bit values)
        yq = y + (y.detach().to(x.dtype).double() - y.detach())
        y = F.max_pool2d(yq, kernel_size=3, stride=2, padding=1)
    out = n["outs"][0]
    errs = {"y": _rel(out, y)}
    assert errs["y"] < tol_h, (kind, tuple(x.shape), activation, errs)
    y.backward(n["gout"][0].double())
    gin = n["gin"]
    if kind == "_AbnAddReluFn":
        dx, dsc, dw, db = gin[:4]
        errs["dshortcut"] = _rel(dsc, sd.grad)
        if errs["dshortcut"] >= tol_h:
            fl = (dsc != 0) != (sd.grad != 0)
            zpre = (_bn_act(x.double(), w.double(), b.double(), eps, activation, slope) + shortcut.double())
            print("DIAG", kind, tuple(x.shape), "mask flips", int(fl.sum()), "of", dsc.numel(), "pre-activation at flips", [round(v, 6) for v in zpre[fl].tolist()[:8]],
                  "y there", out[fl].tolist()[:8], "g there", n["gout"][0][fl].tolist()[:8], "dsc there", dsc[fl].tolist()[:8])
        assert errs["dshortcut"] < tol_h, (kind, tuple(x.shape), errs)
    else:
        dx, dw, db = gin[:3]
    errs["dw"], errs["db"] = _rel(dw, wd.grad), _rel(db, bd.grad)
    if kind == "_IABNFn" and errs["db"] > 1e-3:
        yo, g = n["outs"][0].double(), n["gout"][0].double()
        pre = F.batch_norm(x.double(), None, None, w.double().abs() + eps, b.double(), True, 0.0, eps)
        mism = (yo < 0) != (pre < 0)
        dz_k = torch.where(yo < 0, g * slope, g)
        print("DIAG", kind, tuple(x.shape), "sign(y) != sign(pre):", int(mism.sum()), "of", yo.numel(), "zeros in y:", int((yo == 0).sum()),
              "|pre| at mismatches", [round(v, 8) for v in pre[mism].abs().tolist()[:6]], "db kernel-mask vs kernel:", _rel(db, dz_k.sum((0, 2, 3))),
              "sum|dz|/|sum dz| (worst channel)", float((dz_k.abs().sum((0, 2, 3)) / dz_k.sum((0, 2, 3)).abs()).max()),
              "dy dtype", n["gout"][0].dtype, "max|dy|", float(g.abs().max()), "min nonzero |dy|", float(g.abs()[g != 0].min()))
    if tol_dx is not None:
        errs["dx"] = _rel(dx, xd.grad)
        assert errs["dx"] < tol_dx, (kind, tuple(x.shape), activation, errs)
    assert errs["dw"] < tol_p and errs["db"] < tol_p, (kind, tuple(x.shape), activation, errs)
    return errs


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_every_conv_and_norm_node_of_a_full_step_matches_fp64(dtype, monkeypatch):
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.modeling import ops

    monkeypatch.setenv("MGNET_STREAMS", "0")
    log = []
    for cls in (ops._ConvFn, ops._ShortcutS2Fn, ops._IABNFn, ops._AbnAddReluFn, ops._AbnPoolFn):
        _record(monkeypatch, cls, log)
    cfg, m = small_model(with_depth=True, seed=3)
    _randomise(m)
    m = m.cuda().train()
    m.amp_dtype = dtype
    H, W = 128, 192
    batch = synthetic_batch(2, H, W, "cuda", seed=5)
    losses = m(batch)
    scale = 1.0 if dtype == torch.bfloat16 else 4096.0     # (fp16: a loss scale keeps the small gradients out of the subnormals; 65536 overflows the 1x1 global-context norm of this random net)
    (sum(losses.values()) * scale).backward()
    torch.cuda.synchronize()
    kinds = [n["kind"] for n in log]
    # 2 x ResNet-18 (20 convs each) + 3 decoders (5 each) + heads/predictors + pose decoder; 68 norm sites minus the 7 fused attention norms
    # (the six 1x1 / stride-2 shortcut convs of the two ResNets are _ShortcutS2Fn nodes since round 4)
    assert kinds.count("_ShortcutS2Fn") == 6, {k: kinds.count(k) for k in set(kinds)}
    assert kinds.count("_ConvFn") >= 64 and kinds.count("_AbnAddReluFn") == 16 and kinds.count("_AbnPoolFn") == 2 and kinds.count("_IABNFn") >= 35, \
        {k: kinds.count(k) for k in set(kinds)}
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    seen = {"skip": 0, "skip_lowres": 0, "cout_pad": 0, "stats": 0, "stem": 0, "bias": 0, "tiny_batch_norms": 0}
    worst = {}
    for n in log:
        if n["kind"] == "_ConvFn":
            e = _check_conv(n, dtype, tol_h=1.2 * ulp, tol_f=2e-4 if dtype == torch.bfloat16 else 1e-4)
            seen["skip"] += bool(n["ins"][6])
            seen["cout_pad"] += bool(n["ins"][7])
            seen["stats"] += len(n["ins"]) > 8 and n["ins"][8] is not None
            seen["stem"] += n["ins"][0].shape[1] in (4, 8, 16)
            seen["bias"] += n["ins"][2] is not None
            seen["skip_lowres"] += n["ins"][6] == 2
        elif n["kind"] == "_ShortcutS2Fn":
            e = _check_shortcut_s2(n, dtype, tol_h=1.2 * ulp, tol_f=2e-4 if dtype == torch.bfloat16 else 1e-4)
        else:
            # the norm's result is one 16-bit rounding away from fp64; its backward re-derives x_hat from that rounded output (the
            # in-place contract), which costs the data gradient less than one ulp of a tensor norm (measured: 0.5; bound 3) and the parameter
            # gradients less (they average)
            x = n["ins"][0]
            tiny = x.numel() // x.shape[1] < 8
            seen["tiny_batch_norms"] += tiny
            # (a norm over < 8 samples -- the global-context vector of a 2-frame batch -- has a data gradient that is the small
            #  difference of two projections, (1 - var/(var+eps)) of its terms: x_hat re-derived from a rounded output cannot resolve
            #  it, in this stack as in inplace_abn; its forward and parameter gradients are checked, its dx only for finiteness)
            e = _check_norm(n, tol_h=1.2 * ulp, tol_p=(2.0 if not tiny else 64.0) * ulp,
                            tol_dx=None if tiny else 3.0 * ulp)
            assert all(torch.isfinite(g).all() for g in n["gin"] if torch.is_tensor(g))
        for k, v in e.items():
            worst[(n["kind"], k)] = max(worst.get((n["kind"], k), 0.0), v)
    # every fused configuration named in the docstring did occur in this step
    assert seen["skip"] >= 12 and seen["skip_lowres"] == 6 and seen["cout_pad"] >= 4 and seen["stats"] >= 40 and seen["stem"] == 2 and seen["bias"] >= 4, seen
    print(f"\n[step nodes {str(dtype)[6:]}] {len(log)} nodes; worst relative error per output (one 16-bit ulp = {ulp:.1e}): " +
          ", ".join(f"{a[1:]}.{b} {v:.1e}" for (a, b), v in sorted(worst.items())))
