import os, sys
os.environ["MGN_PLAN_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import critical_path as cp
from mgnet_amd import _C
from test_plan_gpu import _trainer
ta, ba, _ = _trainer(seed=1); tb, bb, _ = _trainer(seed=1)
for k in range(3):
    ta.run_step(ba); tb.run_step(bb)
pa = ta.record_plan(ba); pb = tb.record_plan(bb)
torch.cuda.synchronize()
cp.demangle([it["name"] for it in pa.items])
K = next(k for k, it in enumerate(pa.items) if it["kind"] == 0 and "ins_fwd" in cp.short(it["name"]))
def add(plan):
    it = plan.items[K]
    bufs = []
    for tag, lst in (("W", plan.debug_items[K]["writes"]), ("R", plan.debug_items[K]["reads"])):
        for (a, b) in lst:
            buf = torch.zeros((b - a) // 4, dtype=torch.int32, device="cuda")
            _C.check(_C.lib().mgn_plan_probe(plan.handle, it["node"] | (1 << 24), a, b - a, buf.data_ptr()), "probe")
            bufs.append((tag, b - a, buf))
    return bufs
A, B = add(pa), add(pb)
# the same ranges once more, copied BEHIND THE PREDECESSOR of ins_fwd on its stream (what ins_fwd was about to read)
def add_before(plan):
    it = plan.items[K]
    prev = max(k for k in range(K) if plan.items[k]["kind"] == 0 and plan.items[k]["stream"] == it["stream"])
    bufs = []
    for (a, b) in plan.debug_items[K]["reads"][:2]:
        buf = torch.zeros((b - a) // 4, dtype=torch.int32, device="cuda")
        _C.check(_C.lib().mgn_plan_probe(plan.handle, plan.items[prev]["node"] | (1 << 24), a, b - a, buf.data_ptr()), "probe")
        bufs.append(("before", b - a, buf))
    return bufs
A2, B2 = add_before(pa), add_before(pb)
print("ins_fwd item", K, [(t, n) for t, n, _ in A])
for r in range(600):
    la = {n: float(v) for n, v in ta.replay_plan().items()}
    lb = {n: float(v) for n, v in tb.replay_plan().items()}
    torch.cuda.synchronize()
    if la["loss_offset"] != lb["loss_offset"] or la["loss_center"] != lb["loss_center"]:
        print(f"replay {r}: loss_offset {la['loss_offset']} vs {lb['loss_offset']}")
        for (t, n, x), (_, _, y) in zip(A, B):
            d = (x != y).nonzero().flatten()
            print(f"   {t} {n} bytes: {len(d)} words differ", d[:16].tolist())
            if t == "W" and len(d):
                xf, yf = x.view(torch.float32), y.view(torch.float32)
                for i in d[:8].tolist():
                    print(f"        word {i} (block {i // 4}, quantity {i % 4}): A {float(xf[i]):.9g}  B {float(yf[i]):.9g}")
        for (t, n, x), (_, _, y) in zip(A2, B2):
            d = (x != y).nonzero().flatten()
            print(f"   copied BEFORE ins_fwd, {n} bytes: {len(d)} words differ", d[:16].tolist())
        for (t, n, x), (_, _, y) in zip(A2, A[1:3]):
            print(f"   A: before vs after ins_fwd, {n} bytes: {int((x != y).sum())} words differ;", end="")
        print()
        for (t, n, x), (_, _, y) in zip(B2, B[1:3]):
            print(f"   B: before vs after ins_fwd, {n} bytes: {int((x != y).sum())} words differ;", end="")
        print()
        break
