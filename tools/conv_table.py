#!/usr/bin/env python3
"""Per-signature table of every conv kernel call in ONE MGNet training step at the bench shape: records the calls of a real
step (shapes only), then times each unique signature in isolation.  Output sorted by total time per step."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgnet_amd import _C, add_mgnet_config, get_cfg
from mgnet_amd.data import synthetic_batch
from mgnet_amd.engine import Trainer
from mgnet_amd.registry import build_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, H, W = int(os.environ.get("B", 8)), int(os.environ.get("H", 1024)), int(os.environ.get("W", 2048))
dev = torch.device("cuda:0")
cfg = get_cfg(); add_mgnet_config(cfg)
cfg.merge_from_file(os.path.join(ROOT, "configs", "bench-c4-cityscapes-videosequence.yaml"))
cfg.merge_from_list(["MODEL.DEVICE", str(dev), "SOLVER.IMS_PER_BATCH", B, "MODEL.SEM_SEG_HEAD.OHEM_N_MIN", min(524287, B * H * W // 4 - 1)])
torch.manual_seed(0)
model = build_model(cfg); trainer = Trainer(cfg, model)
batch = synthetic_batch(B, H, W, dev, seed=1234)
trainer.run_step(batch)

calls = collections.Counter()
o_ig, o_wg = _C.conv_igemm, _C.conv_wgrad

def r_ig(x, w, out_shape, bias, stride, pad, up=1, relu=False, out_dtype=torch.bfloat16, khw=None, residual=None, stats=None):
    calls[("igemm", tuple(x.shape), tuple(w.shape), tuple(out_shape), stride, pad, up, khw, out_dtype)] += 1
    return o_ig(x, w, out_shape, bias, stride, pad, up, relu, out_dtype, khw, residual, stats)

def r_wg(dy, x, kh, kw, stride, pad, cin_real=None, lazy=False):
    calls[("wgrad", tuple(dy.shape), tuple(x.shape), kh, kw, stride, pad, cin_real)] += 1
    return o_wg(dy, x, kh, kw, stride, pad, cin_real, lazy=lazy)

o_up2 = _C.conv_up2

def r_up2(dy, w, out_hw, residual=None, residual_lowres=False):   # (the stride-2 data gradients with the shortcut's gradient at low resolution)
    calls[("up2", tuple(dy.shape), tuple(w.shape), tuple(out_hw), bool(residual_lowres))] += 1
    return o_up2(dy, w, out_hw, residual, residual_lowres)

_C.conv_igemm, _C.conv_wgrad, _C.conv_up2 = r_ig, r_wg, r_up2
trainer.run_step(batch)
_C.conv_igemm, _C.conv_wgrad, _C.conv_up2 = o_ig, o_wg, o_up2
torch.cuda.synchronize()

def timeit(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

def cl(shape):
    return torch.randn(*shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

rows = []
for sig, cnt in calls.items():
    if sig[0] == "igemm":
        _, xs, ws, osz, s, p, up, khw, odt = sig
        x = cl(xs); w = (torch.randn(*ws, device=dev) * 0.05).to(torch.bfloat16)
        kh, kw = khw if khw else ws[1:3]
        gf = 2.0 * xs[0] * osz[0] * osz[1] * ws[0] * xs[1] * kh * kw / 1e9 / (up * up)
        # the stems run on channel-padded input (3 -> 8, 9 -> 16: csrc/prep.hip): the padding is not work
        real = gf * ({8: 3, 16: 9}.get(xs[1], xs[1]) / xs[1] if (kh, kw) == (7, 7) else 1.0)
        t = timeit(lambda: o_ig(x, w, osz, None, s, p, up, False, odt, khw))
        desc = f"igemm x{xs} w{ws} out{osz} s{s} p{p} up{up}{' packed' if khw else ''}{' f32' if odt == torch.float32 else ''}"
    elif sig[0] == "up2":
        _, dys, ws, osz, lo = sig
        dy = cl(dys); w = (torch.randn(*ws, device=dev) * 0.05).to(torch.bfloat16)
        res = cl((dys[0], ws[0], dys[2], dys[3])) if lo else None
        gf = real = 2.0 * dys[0] * dys[2] * dys[3] * dys[1] * ws[0] * ws[1] * ws[2] / 1e9
        t = timeit(lambda: o_up2(dy, w, osz, res, lo))
        desc = f"up2   dy{dys} w{ws} out{osz} s2 data gradient (conv_up2.hip){' + low-res shortcut gradient' if lo else ''}"
    else:
        _, dys, xs, kh, kw, s, p, cr = sig
        dy = cl(dys); x = cl(xs)
        gf = 2.0 * dys[0] * dys[2] * dys[3] * dys[1] * xs[1] * kh * kw / 1e9
        real = gf * ((cr / xs[1]) if cr else 1.0)
        t = timeit(lambda: o_wg(dy, x, kh, kw, s, p, cr))
        desc = f"wgrad dy{dys} x{xs} k{kh}x{kw} s{s} p{p} cin_real={cr}"
    rows.append((cnt * t, cnt, t, gf, real, desc))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
tf_issued, tf_real = sum(r[1] * r[3] for r in rows) / 1e3, sum(r[1] * r[4] for r in rows) / 1e3
print(f"total conv time per step: {tot:.2f} ms, {sum(r[1] for r in rows)} calls, {tf_real:.2f} TFLOP of real work = {tf_real / tot * 1e3:.0f} TF/s = "
      f"{tf_real / tot * 1e3 / 2500 * 100:.1f} % of the 2.5 PF dense bf16 peak ({tf_issued:.2f} TFLOP issued incl. the stems' channel padding)")
print("   total      calls x each      issued GF  TF/s | real GF  TF/s  signature")
for tt, cnt, t, gf, real, desc in rows:
    print(f"{tt:7.3f} ms  {cnt:2d} x {t*1e3:7.1f} us  {gf:7.1f} GF {gf/t:6.0f} | {real:7.1f} {real/t:6.0f}  {desc}")
