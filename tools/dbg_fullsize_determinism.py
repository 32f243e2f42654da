"""Where does bit-reproducibility end at the benchmark's size (8 frames of 1024x2048)?  From one state snapshot, each of the following is run
several times and compared bit for bit (losses of every step, final gradient buckets, final parameters):
  eager    : K eager training steps
  plan     : K replays of ONE recording
  plans    : K replays of recording A vs K replays of recording B
Usage: dbg_fullsize_determinism.py [BxHxW] [K] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mgnet_amd import add_mgnet_config, get_cfg
from mgnet_amd.data import synthetic_batch
from mgnet_amd.engine import Trainer
from mgnet_amd.registry import build_model

B, H, W = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "8x1024x2048").split("x")]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
cfg = get_cfg(); add_mgnet_config(cfg)
cfg.merge_from_file(os.path.join(ROOT, "configs", "bench-c4-cityscapes-videosequence.yaml"))
cfg.merge_from_list(["MODEL.DEVICE", str(dev), "SOLVER.IMS_PER_BATCH", B, "MODEL.SEM_SEG_HEAD.OHEM_N_MIN", min(524287, B * H * W // 4 - 1)])
torch.manual_seed(0)
tr = Trainer(cfg, build_model(cfg))
batch = synthetic_batch(B, H, W, dev, seed=1234)
for _ in range(3):
    tr.run_step(batch)
torch.cuda.synchronize()
snap = tr.state_snapshot()


SUMS = []   # per run: [step][bucket] checksum of the gradient bucket after the step (stream-ordered, no synchronisation)


def run(step_fn):
    tr.state_restore(snap)
    ls, cs = [], []
    for _ in range(K):
        ld = step_fn()
        ls.append(torch.stack([v.detach().float().reshape(()) for v in ld.values()]).clone())
        cs.append(torch.stack([b["flat_g"].view(torch.int32).to(torch.int64).sum() for b in tr.reducer.buckets]))
        if os.environ.get("SYNC_EACH") == "1":   # (the host never runs ahead of the device by more than one step)
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return (torch.stack(ls).cpu(), [b["flat_g"].clone() for b in tr.reducer.buckets], [p.detach().clone() for p in tr.model.parameters()],
            torch.stack(cs).cpu())


def compare(a, b, names):
    la, ga, pa, sa_ = a
    lb, gb, pb, sb_ = b
    same = la.view(torch.int32) == lb.view(torch.int32)
    out = []
    if not bool(same.all()):
        k = int((~same.all(dim=1)).nonzero()[0])
        out.append(f"losses differ first at step {k}: " + ", ".join(f"{names[j]} {float(la[k, j]):.9g}/{float(lb[k, j]):.9g}" for j in range(len(names)) if not bool(same[k, j])))
    ng = sum(not torch.equal(x, y) for x, y in zip(ga, gb))
    npar = sum(not torch.equal(x, y) for x, y in zip(pa, pb))
    if ng or npar:
        out.append(f"{ng}/{len(ga)} gradient buckets, {npar}/{len(pa)} parameters differ")
        if npar:
            nm = [n for n, _ in tr.model.named_parameters()]
            bad = [nm[i] for i, (x, y) in enumerate(zip(pa, pb)) if not torch.equal(x, y)]
            out.append("   first / last differing parameters: " + ", ".join(bad[:4]) + " ... " + ", ".join(bad[-4:]))
    if not torch.equal(sa_, sb_):
        out.append("gradient buckets differing as [step, bucket]: " + str((sa_ != sb_).nonzero().tolist()[:12]))
    return out


names = list(tr.run_step(batch).keys())
modes = os.environ.get("MODES", "eager,plan,plans").split(",")
if "eager" in modes:
    ref = run(lambda: tr.run_step(batch))
    for r in range(REPS):
        d = compare(ref, run(lambda: tr.run_step(batch)), names)
        print(f"[eager rep {r}] " + ("identical" if not d else " | ".join(d)), flush=True)
if "plan" in modes or "plans" in modes:
    tr.state_restore(snap)
    pa = tr._record_plan_once(batch)
    sa = (tr._plan, tr._plan_losses, tr._plan_inputs, tr._plan_keep)
    pb = tr._record_plan_once(batch)
    sb = (tr._plan, tr._plan_losses, tr._plan_inputs, tr._plan_keep)

    def use(st):
        tr._plan, tr._plan_losses, tr._plan_inputs, tr._plan_keep = st
    if "plan" in modes:
        for nm, st in (("A", sa), ("B", sb)):
            use(st)
            ref = run(lambda: tr.replay_plan())
            for r in range(REPS):
                d = compare(ref, run(lambda: tr.replay_plan()), names)
                print(f"[plan {nm} vs itself, rep {r}] " + ("identical" if not d else " | ".join(d)), flush=True)
    if "plans" in modes:
        for r in range(REPS):
            use(sa)
            ra = run(lambda: tr.replay_plan())
            use(sb)
            rb = run(lambda: tr.replay_plan())
            d = compare(ra, rb, names)
            print(f"[plan A vs plan B, rep {r}] " + ("identical" if not d else " | ".join(d)), flush=True)
        use(sa)
        ra = run(lambda: tr.replay_plan())
        re = run(lambda: tr.run_step(batch))
        d = compare(ra, re, names)
        print("[plan A vs eager] " + ("identical" if not d else " | ".join(d)), flush=True)
