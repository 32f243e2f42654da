import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import race_screen as rs
from mgnet_amd import _C
dev = rs.dev
side = [torch.cuda.Stream() for _ in range(2)]
big = torch.randn(64 << 20, device=dev)
mm = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
want = sys.argv[1:] or ["conv1x1 256->256 @128x256", "conv3x3 s1 64->64 @256x512", "stem 7x7 s2 3(4)->64"]
BUSY = os.environ.get("BUSY", "mm,mul")
for name, fn, _ref in rs.cases(8):
    if name not in want:
        continue
    ref = fn(); torch.cuda.synchronize()
    for r in range(4):
        for st in side:
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                if "mul" in BUSY: big.mul_(1.0001)
                if "mm" in BUSY: torch.mm(mm, mm)
        y = fn()
        torch.cuda.synchronize()
        idx = (y != ref).nonzero()
        print(f"{name} rep {r}: {idx.shape[0]} differ", flush=True)
        for i in idx[:6].tolist():
            a, b = ref[tuple(i)], y[tuple(i)]
            print(f"     at {i}: first {float(a):.6f} ({a.view(torch.int16).item() & 0xffff:#06x}) now {float(b):.6f} ({b.view(torch.int16).item() & 0xffff:#06x})")
    # and with NO side work
    nb = 0
    for r in range(6):
        y = fn(); torch.cuda.synchronize()
        nb += int((y != ref).sum() > 0)
    print(f"{name}: without side streams {nb} of 6 launches differ")
