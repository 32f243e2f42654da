"""Which launch of the recorded step is the FIRST whose output differs between two replays from the same state?  Behind every launch a
checksum of every block it writes is taken on its own stream (csrc/plan.hip mgn_plan_probe); the replays are compared launch by launch in
issue order.  A launch whose outputs differ while the outputs of everything it reads are equal is where a replay went wrong.
usage: dbg_plan_first_diff.py [BxHxW] [replays]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["MGN_PLAN_DEBUG"] = "1"
import argparse
import numpy as np
import torch
import critical_path as cp
from mgnet_amd import _C

B, H, W = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "8x1024x2048").split("x")]
R = int(sys.argv[2]) if len(sys.argv) > 2 else 12
args = argparse.Namespace(batch=B, height=H, width=W, dtype="bf16")
dev = torch.device("cuda", 0)
tr, batch = cp.build_trainer(args, dev)
for _ in range(4):
    tr.run_step(batch)
snap = tr.state_snapshot()
plan = tr.record_plan(batch)
items, dbg = plan.items, plan.debug_items
cp.demangle([it["name"] for it in items])
MAXB = int(os.environ.get("PROBE_MAX_MB", 600)) << 20
probes = []   # (item index, range index, a, b)
# shared workspaces (a block many launches write: the per-stream norm workspace, ...) keep stale bytes of earlier launches in the parts a
# launch does not rewrite -- their checksums differ after ANY earlier difference and say nothing: skipped
nwriters = {}
for it in dbg:
    if it["kind"] == 0:
        for r_ in set(it["writes"]):
            nwriters[r_] = nwriters.get(r_, 0) + 1
MAXW = int(os.environ.get("PROBE_MAX_WRITERS", 4))
for k, it in enumerate(dbg):
    if it["kind"] != 0:
        continue
    seen = set()
    for (a, b) in it["writes"]:
        if (a, b) in seen or b - a <= 1 or b - a > MAXB or nwriters.get((a, b), 0) > MAXW:
            continue
        seen.add((a, b))
        probes.append((k, a, b - (b - a) % 4))
# FOCUS=<kernel name substring>: additionally, behind the launch that PRECEDES the first such kernel on its stream, checksum everything that
# kernel reads and writes -- "were its inputs still intact when it started?"
focus = os.environ.get("FOCUS")
nprobe_out = len(probes)
if focus:
    kf = next(k for k, it in enumerate(items) if it["kind"] == 0 and focus in cp.short(it["name"]))
    # (the previous LAUNCH on its stream: probes behind a prof mark are never run by mgn_plan_run)
    prev = max(k for k in range(kf) if items[k]["kind"] == 0 and items[k]["stream"] == items[kf]["stream"] and "prof" not in items[k]["name"])
    seen = set()
    for (a, b) in dbg[kf]["reads"] + dbg[kf]["writes"]:
        if (a, b) not in seen and 1 < b - a <= MAXB:
            seen.add((a, b))
            probes.append((prev, a, b - (b - a) % 4))
    npre = len(probes)
    for (a, b) in dbg[kf]["reads"]:      # ... and once more BEHIND it: was an input rewritten while it ran?
        if 1 < b - a <= MAXB:
            probes.append((kf, a, b - (b - a) % 4))
    print(f"focus: #{kf} {cp.short(items[kf]['name'])[:40]}: {len(probes) - nprobe_out} input / output ranges probed behind #{prev} {cp.short(items[prev]['name'])[:30]}")
slots = torch.zeros(len(probes), dtype=torch.int64, device=dev)
# FOCUS_COPY=1: the focus kernel's written ranges are also COPIED behind it every replay, to see WHERE in them two replays differ
copies = []
if focus and os.environ.get("FOCUS_COPY") == "1":
    seen = set()
    for (a, b) in dbg[kf]["writes"]:
        if (a, b) not in seen and 1 < b - a <= MAXB:
            seen.add((a, b))
            buf = torch.zeros((b - a) // 4, dtype=torch.int32, device=dev)
            copies.append((a, b, buf))
lib = _C.lib()
for j, (k, a, b) in enumerate(probes):
    _C.check(lib.mgn_plan_probe(plan.handle, items[k]["node"], ctypes.c_void_p(a), b - a, ctypes.c_void_p(slots.data_ptr() + 8 * j)), "probe")
for (a, b, buf) in copies:
    _C.check(lib.mgn_plan_probe(plan.handle, items[kf]["node"] | (1 << 24), ctypes.c_void_p(a), (b - a) // 4 * 4, ctypes.c_void_p(buf.data_ptr())), "probe copy")
print(f"{len(probes)} probes behind {len({p[0] for p in probes})} launches, {sum(b - a for _, a, b in probes) / 1e9:.2f} GB checksummed per replay", flush=True)
sid = {st: i for i, st in enumerate(sorted({it['stream'] for it in items}))}
res = []
for r in range(R):
    tr.state_restore(snap)
    slots.zero_()
    torch.cuda.synchronize()
    ld = tr.replay_plan()
    torch.cuda.synchronize()
    res.append((slots.cpu().numpy().copy(), {k: float(v) for k, v in ld.items()}))
    if copies:
        if r == 0:
            ref_copies = [buf.clone() for (_a, _b, buf) in copies]
        else:
            for (a, b, buf), rc in zip(copies, ref_copies):
                d = (buf != rc).nonzero().flatten()
                if d.numel():
                    off = d * 4
                    # which other items touch memory right around the differing bytes?
                    lo_, hi_ = a + int(off[0]), a + int(off[-1]) + 4
                    near = []
                    for k2, it2 in enumerate(dbg):
                        for (x, y) in it2["writes"]:
                            if k2 != kf and (abs(y - lo_) < (1 << 20) or abs(x - hi_) < (1 << 20) or (x < hi_ and y > lo_)):
                                near.append(f"#{k2}[s{sid[items[k2]['stream']]}]{cp.short(items[k2]['name'])[:22]}({x - a:+d}..{y - a:+d})")
                                break
                    vals = [(int(o), int(rc[o // 4 if False else int(o) // 4]), int(buf[int(o) // 4])) for o in off[:4].tolist()]
                    print(f"[replay {r}] COPY of {a:#x}+{(b - a) >> 10} KB behind the focus kernel: {d.numel()} words differ, byte offsets {int(off[0])} .. {int(off[-1])} "
                          f"(first words ref/now: {[(o, hex(x & 0xffffffff), hex(y & 0xffffffff)) for o, x, y in vals]}); writers of nearby memory: {' '.join(near[:10])}", flush=True)
ref = res[0]
for r in range(1, R):
    d = np.nonzero(res[r][0] != ref[0])[0]
    dl = {k: (ref[1][k], v) for k, v in res[r][1].items() if v != ref[1][k]}
    if len(d) == 0:
        print(f"[replay {r}] identical (losses differing: {dl})")
        continue
    first = [probes[j] for j in d[:6]]
    if focus:
        fd = [j for j in d if nprobe_out <= j < npre]
        print(f"[replay {r}] focus ranges differing BEFORE the kernel started: " + (", ".join(f"{probes[j][1]:#x}+{(probes[j][2] - probes[j][1]) >> 10} KB" for j in fd) or "none"))
        fa = [j for j in d if j >= npre]
        print(f"[replay {r}] focus INPUT ranges differing right AFTER it: " + (", ".join(f"{probes[j][1]:#x}+{(probes[j][2] - probes[j][1]) >> 10} KB" for j in fa) or "none"))
        fd = fd + fa
        for j in fd:
            a, b = probes[j][1], probes[j][2]
            wr = [k for k, it in enumerate(dbg) if any(x < b and y > a for x, y in it["writes"])]
            rd = [k for k, it in enumerate(dbg) if any(x < b and y > a for x, y in it["reads"])]
            desc = lambda k: f"#{k}[s{sid[items[k]['stream']]}]{'T:' if items[k]['kind'] else ''}{cp.short(items[k]['name'])[:24]}"
            print(f"      range {a:#x}: writers " + " ".join(desc(k) for k in wr[:14]) + " | readers " + " ".join(desc(k) for k in rd[:14]))
    print(f"[replay {r}] {len(d)} of {len(probes)} checksums differ; losses differing: {dl}; first differing launches in issue order: " +
          "; ".join(f"#{k} [s{sid[items[k]['stream']]}] {cp.short(items[k]['name'])[:36]} ({(b - a) >> 10} KB)" for k, a, b in first), flush=True)
