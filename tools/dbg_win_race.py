"""Race screen of the windowed 3x3 kernels (csrc/conv_win.hip) at the shapes of the C4 step: the same convolution launched repeatedly
-- alternately with and without the statistics epilogue, other streams busy beside it -- must give the same bits every time.
`MGNET_HIP_LIB=<other .so>` selects a build variant.  Usage: dbg_win_race.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgnet_amd import _C

dev = torch.device("cuda:0")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
SHAPES = [(8, 128, 128, 256, 128, 8), (8, 256, 128, 256, 256, 8), (8, 128, 64, 128, 128, 8), (8, 128, 32, 64, 256, 16), (8, 512, 32, 64, 512, 16), (8, 256, 64, 128, 256, 8)]
g = torch.Generator(device="cuda").manual_seed(3)
side = [torch.cuda.Stream() for _ in range(2)]
big = torch.randn(64 << 20, device=dev)
mm = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
total_bad = 0
for (N, Cin, H, W, Cout, pr) in SHAPES:
    x = torch.randn(N, Cin, H, W, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev, generator=g) * 0.05).to(torch.bfloat16).contiguous()
    shift = torch.zeros(Cout, device=dev)
    ref = _C.conv3x3_win(x, w, patch_rows=pr)
    torch.cuda.synchronize()
    # fp32 reference of a few channels (is the FIRST result right?)
    xr = x[:1].float()
    wr = w[:8].float().permute(0, 3, 1, 2).contiguous()
    want = torch.nn.functional.conv2d(xr, wr, padding=1)
    err = float((ref[:1, :8].float() - want).abs().max() / want.abs().max())
    bad = []
    for r in range(REPS):
        if os.environ.get("BUSY", "1") == "1":
            for st in side:
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    big.mul_(1.0001)
                    torch.mm(mm, mm)
        if r % 2:
            y, part = _C.conv3x3_win(x, w, patch_rows=pr, stats_shift=shift, want_stats=True)
        else:
            y = _C.conv3x3_win(x, w, patch_rows=pr)
        n = int((y != ref).sum())
        if n:
            d = (y.float() - ref.float()).abs()
            idx = (y != ref).nonzero()
            bad.append((r, "stats" if r % 2 else "plain", n, float(d.max()), idx[0].tolist(), idx[-1].tolist()))
    total_bad += len(bad)
    print(f"x({N},{Cin},{H},{W}) -> {Cout}, patch rows {pr}: first result vs fp32 {err:.2e}; {len(bad)} of {REPS} launches differ from the first" +
          ("" if not bad else ": " + "; ".join(f"#{r} {k}: {n} elements, max |d| {d:.2f}, first {a} last {b}" for r, k, n, d, a, b in bad[:4])), flush=True)
print(f"launches that differed: {total_bad}   lib={os.environ.get('MGNET_HIP_LIB', 'in-tree')}")
