import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from test_plan_gpu import _trainer
ta, ba, _ = _trainer(seed=1); tb, bb, _ = _trainer(seed=1)
for k in range(3):
    ta.run_step(ba); tb.run_step(bb)
EAGER = os.environ.get("EAGER") == "1"
if not EAGER:
    pa = ta.record_plan(ba); pb = tb.record_plan(bb)
torch.cuda.synchronize()
if os.environ.get("FENCE") and not EAGER:
    import critical_path as cp
    from mgnet_amd import _C
    for plan in (pa, pb):
        cp.demangle([it["name"] for it in plan.items])
        n = 0
        for it in plan.items:
            if it["kind"] == 0 and any(x in cp.short(it["name"]) for x in os.environ["FENCE"].split(",")):
                _C.check(_C.lib().mgn_plan_set_skip(plan.handle, it["node"], 3), "fence"); n += 1
    print("fences behind", n, "nodes")
def grads(t):
    return torch.cat([b["flat_g"].reshape(-1) for b in t.reducer.buckets]).clone()
nbad = 0
BG = os.environ.get("BG") == "1"
if BG:
    bgs = torch.cuda.Stream()
    m1 = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16); m2 = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    x1 = torch.randn(64 << 20, device="cuda")
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    if BG:   # unrelated work on another stream while B replays: MFMA-heavy and streaming kernels sharing the CUs
        with torch.cuda.stream(bgs):
            for _ in range(6):
                torch.mm(m1, m2); x1.mul_(1.0001)
    if EAGER:
        la = {n: float(v) for n, v in ta.run_step(ba).items()}
        lb = {n: float(v) for n, v in tb.run_step(bb).items()}
    else:
        pb.set_jitter(1000 + k, permille=int(os.environ.get("JP", 150)), max_us=int(os.environ.get("JU", 300)))
        la = {n: float(v) for n, v in ta.replay_plan().items()}
        lb = {n: float(v) for n, v in tb.replay_plan().items()}
    torch.cuda.synchronize()
    ga, gb = grads(ta), grads(tb)
    if la != lb or not torch.equal(ga, gb):
        nbad += 1
        print(f"replay {k}: DIFFERS losses {[(n, la[n], lb[n]) for n in la if la[n] != lb[n]]} grads max diff {float((ga - gb).abs().max()):.3e}")
        names = []
        for (x, y) in zip(ta.reducer.buckets, tb.reducer.buckets):
            for p, o in zip(y["params"], y["offsets"]):
                if not torch.equal(x["flat_g"][o:o + p.numel()], y["flat_g"][o:o + p.numel()]):
                    names.append([n for n, q in tb.model.named_parameters() if q is p][0])
        print("      differing grads:", len(names), names[:12], "...", names[-4:])
        # re-align B with A: copy parameters, buffers and optimizer state so that later replays are compared step by step
        with torch.no_grad():
            for x, y in zip(ta.reducer.buckets, tb.reducer.buckets):
                y["flat_p"].copy_(x["flat_p"])
            for (na, ba_), (nb, bb_) in zip(ta.model.named_buffers(), tb.model.named_buffers()):
                bb_.copy_(ba_)
            for k_, v in ta.optimizer.__dict__.items():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    getattr(tb.optimizer, k_).copy_(v)
                elif isinstance(v, list) and v and isinstance(v[0], torch.Tensor):
                    for u, w in zip(v, getattr(tb.optimizer, k_)):
                        w.copy_(u)
print("replays with jitter that differed:", nbad)
