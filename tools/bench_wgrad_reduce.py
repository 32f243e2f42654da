#!/usr/bin/env python3
"""The batched split-K reduction of the weight gradients (`conv_wgrad_reduce_batch`, one launch per gradient bucket; 6 per C4 step):
µs, bytes of partial tiles read and the bandwidth it reaches, per group of layers as the gradient reducer hands them over.
`MGNET_HIP_LIB=<other .so>` for an A/B on the same box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgnet_amd import _C

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 8))
# (Cin, Cout, H, W, k, stride) groups resembling the buckets of the C4 step (decoder / head layers, res5+res4, res3+res2)
GROUPS = {
    "heads 256->256 @128x256 x4": [(256, 256, 128, 256, 3, 1)] * 4,
    "decoders 128->128 @128x256 x9": [(128, 128, 128, 256, 3, 1)] * 9,
    "res5 512->512 @32x64 x6 + res4 256->256 @64x128 x6": [(512, 512, 32, 64, 3, 1)] * 6 + [(256, 256, 64, 128, 3, 1)] * 6,
    "res3 128->128 @128x256 x6 + res2 64->64 @256x512 x8": [(128, 128, 128, 256, 3, 1)] * 6 + [(64, 64, 256, 512, 3, 1)] * 8,
}


def cl(*shape):
    return torch.randn(*shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


tot_us = tot_b = 0.0
for name, layers in GROUPS.items():
    _C.WGRAD_LAZY[0] = 4711
    entries, keep = [], []
    cache = {}
    for (Cin, Cout, H, W, k, s) in layers:
        key = (Cin, Cout, H, W)
        if key not in cache:
            cache[key] = (cl(B, Cin, H, W), cl(B, Cout, H // s, W // s))
        x, dy = cache[key]
        dw = _C.conv_wgrad(dy, x, k, k, s, k // 2, lazy=True)
        desc, ws, shape, _ = _C.WGRAD_PENDING.pop(dw.data_ptr())
        entries.append((desc, ws, dw))
        keep.append(dw)
    _C.WGRAD_LAZY[0] = False
    nbytes = sum(d[2] * d[3] * d[4] * d[5] * 4 for d, _w, _o in entries)
    # reference: the plain sum over the splits in fp64
    _C.wgrad_reduce_batch(entries)
    torch.cuda.synchronize()
    d, ws, dw = entries[0]
    part = ws.view(torch.float32)[: d[2] * d[3] * d[4] * d[5]].view(d[2], d[3], d[4], d[5]).double().sum(0)   # [Cout][taps][Cin]
    ref = part.view(d[3], int(d[4] ** 0.5), int(d[4] ** 0.5), d[5]).permute(0, 3, 1, 2)
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(3):
        _C.wgrad_reduce_batch(entries)
    torch.cuda.synchronize()
    n = 20
    ev[0].record()
    for _ in range(n):
        _C.wgrad_reduce_batch(entries)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) / n * 1e3
    tot_us += us
    tot_b += nbytes
    print(f"{name:58s} splits {sorted({d[2] for d, _w, _o in entries})}  {nbytes / 1e6:7.1f} MB  {us:7.1f} us  {nbytes / us * 1e-6:5.2f} TB/s")
print(f"sum: {tot_us / 1e3:.3f} ms for {tot_b / 1e9:.2f} GB = {tot_b / tot_us * 1e-6:.2f} TB/s   lib={os.environ.get('MGNET_HIP_LIB', 'in-tree')}")
