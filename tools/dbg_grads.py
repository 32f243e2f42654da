import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["MGNET_ALLOW_TORCH_STAGING"] = "1"
import torch
from test_grad_parity_gpu import _grads
H, W, amp, wd = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == "bf16", sys.argv[4] == "depth"
ref, got, rows = _grads(H, W, amp=amp, with_depth=wd)
print({k: (round(float(got[k]), 5), round(float(ref[k]), 5)) for k in ref})
tot = sum(r[3] ** 2 for r in rows) ** 0.5
rows.sort(key=lambda r: r[1])
for n, c, e, rn in rows[:25]:
    print(f"{n:60s} cos {c:8.5f} rel {e:8.4f} |g|/tot {rn / tot:9.2e}")
print("median cos", sorted(r[1] for r in rows)[len(rows) // 2], "n", len(rows), "n cos<0.999", sum(r[1] < 0.999 for r in rows))
