#!/usr/bin/env python3
"""Where the training step's time goes, measured INSIDE un-profiled launch-plan replays (no rocprofv3: under the profiler the step is
host-bound and its idle share says nothing about the real run).

The step measured is the benchmark's (bench.py: 8 frames of 1024x2048 per GPU, bf16); the loop it belongs to is
tools/train_net.py:232-234 of the reference.  Three measurements, all on the recorded plan (engine/plan.py):

 (a) TRACE -- `mgn_plan_trace`: every launch of a replay carries a hipEvent pair bound to its own dispatch (hipExtLaunchKernel), which
     gives begin / end of each of the step's ~700 kernels on the device clock without marker packets in the queues.  Per stream: busy /
     idle, where the idle time sits (which kernel pairs), concurrency; per kernel family: time in the step.
 (b) CRITICAL PATH -- the schedule's dependency edges (same-stream order + the cross-stream events of derive_schedule) walked backwards
     from the kernel that ends last, always to the predecessor that ended last: the chain of kernels and hand-over gaps that determines the
     step time.  Time on the chain by family = what shortening that family can buy at most.
 (c) WHAT-IF -- `mgn_plan_set_skip`: replays with one family's launches left out (results are garbage, timing is not): the step time
     without that family = what the family costs on the critical path AND through contention.  The optimizer launches are skipped in all
     what-if runs so that the parameters stay finite.

usage: python tools/critical_path.py [--out gpurun_out/critical_path] [--steps 20] [--families name=regex,...]
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FAMILIES = [   # first match wins
    ("stem fwd", r"conv_stem7"),
    ("stem wgrad", r"conv_wgrad_stem"),
    ("stem pool", r"abn_maxpool"),
    ("wgrad 3x3", r"conv_wgrad3x3"),
    ("wgrad reduce", r"conv_wgrad_reduce"),
    ("wgrad other", r"conv_wgrad"),
    ("conv win (3x3 s1 fwd+dgrad)", r"conv3x3_win"),
    ("conv c64", r"conv3x3_c64"),
    ("conv up2 (s2 dgrad)", r"up2_win"),
    ("conv 1x1 stream", r"conv1x1_s_"),
    ("conv igemm (s2 fwd, 1x1, other)", r"conv_igemm"),
    ("iabn bwd reduce", r"iabn_bwd_reduce|colsum_partial"),
    ("iabn bwd apply", r"iabn_bwd_apply"),
    ("iabn apply / add_relu fwd", r"iabn_apply|abn_add_relu"),
    ("iabn coeffs / stats", r"iabn_from_partials|iabn_stats|iabn_partials|iabn_coeff|colsum_final"),
    ("reproj", r"reproj_"),
    ("head losses", r"upce_|ohem_|ins_fwd|ins_bwd|uncertainty|head_act|sum3_kernel|sum4_kernel"),
    ("upsample depth", r"up1_|adjoint_gather"),
    ("attention / small vec (+ fused norm passes)", r"vec_linear|vec_sum|scale_channels|bcast_rows|att_abn|abn_apply_pool"),
    ("eltwise (nearest, concat, sum3, relu)", r"nearest_|concat2|split2|sum3_h16|relu_mask|add_relu"),
    ("input prep / weight layout", r"prep_kernel|u8_frames|weight_layout|copy_from_host"),
    ("optimizer", r"adam_kernel|sqnorm|clip_coef|optim"),
]


def family_of(name, fams):
    name = _DEMANGLED.get(name, name)
    for f, rx in fams:
        if re.search(rx, name):
            return f
    return "other"


_DEMANGLED = {}


def demangle(names):
    """kernel names come back mangled from the runtime (hipKernelNameRefByPtr): one c++filt call for all of them"""
    import subprocess
    todo = sorted({n for n in names if n.startswith("_Z") and n not in _DEMANGLED})
    if todo:
        try:
            r = subprocess.run(["c++filt"], input="\n".join(todo), capture_output=True, text=True, timeout=60)
            for a, b in zip(todo, r.stdout.splitlines()):
                _DEMANGLED[a] = b
        except Exception:  # noqa: BLE001
            pass
    return [_DEMANGLED.get(n, n) for n in names]


def short(name):
    name = _DEMANGLED.get(name, name)
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:44]


def build_trainer(args, dev):
    from mgnet_amd import add_mgnet_config, get_cfg
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer
    from mgnet_amd.registry import build_model
    cfg = get_cfg()
    add_mgnet_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "bench-c4-cityscapes-videosequence.yaml"))
    B, H, W = args.batch, args.height, args.width
    cfg.merge_from_list(["MODEL.DEVICE", str(dev), "SOLVER.IMS_PER_BATCH", B, "MODEL.SEM_SEG_HEAD.OHEM_N_MIN", min(524287, B * H * W // 4 - 1),
                         "SOLVER.AMP.DTYPE", {"bf16": "bfloat16", "fp16": "float16"}[args.dtype]])
    torch.manual_seed(0)
    trainer = Trainer(cfg, build_model(cfg))
    batch = synthetic_batch(B, H, W, dev, seed=1234)
    return trainer, batch


def timed(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def schedule_edges(plan):
    """-> per item index: list of predecessor item indices (previous item of its stream + the items whose events it waits for)"""
    from mgnet_amd.engine.plan import BREAK, LAUNCH, RECORD, WAIT
    ops, items = plan.ops, plan.items
    preds = [[] for _ in items]
    last_on = {}           # stream -> item index
    ev_src = {}            # event -> item index that precedes its record (or None: start of step)
    pending = {}           # stream -> [item indices waited for]
    k = -1                 # item counter: LAUNCH and BREAK ops appear in item order
    for t, a, st in ops:
        if t == RECORD:
            ev_src[a] = last_on.get(st)
        elif t == WAIT:
            src = ev_src.get(a)
            if src is not None:
                pending.setdefault(st, []).append(src)
        else:
            k += 1
            assert items[k]["kind"] == (0 if t == LAUNCH else 1) and items[k]["stream"] == st
            if st in last_on:
                preds[k].append(last_on[st])
            preds[k].extend(pending.pop(st, []))
            last_on[st] = k
    assert k == len(items) - 1
    return preds


def union_len(iv):
    iv = sorted(iv)
    tot, cur_a, cur_b = 0.0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        tot += cur_b - cur_a
    return tot


def analyse(plan, begins, ends, fams, out):
    """begins / ends: [steps][nodes] ms.  Writes the report lines into `out` and returns a dict of the headline figures."""
    items = plan.items
    demangle([it["name"] for it in items])
    preds = schedule_edges(plan)
    node_of = np.array([it["node"] for it in items])
    kern = np.array([it["kind"] == 0 for it in items])
    S = len(begins)
    # per item times: median over the traced steps
    b = np.full((S, len(items)), np.nan)
    e = np.full((S, len(items)), np.nan)
    for s in range(S):
        b[s, kern] = begins[s][node_of[kern]]
        e[s, kern] = ends[s][node_of[kern]]
    ok = kern & ~np.isnan(e).any(0)
    dur = np.nanmedian(e - b, 0)
    span = np.nanmax(e, 1) - np.nanmin(b, 1)
    res = {"trace_span_ms": float(np.median(span)), "kernels": int(ok.sum())}
    out.append(f"traced kernels per step: {int(ok.sum())}; span first begin -> last end: median {np.median(span):.3f} ms "
               f"(min {span.min():.3f}, max {span.max():.3f}) over {S} traced replays")
    # ---- (a) per stream ------------------------------------------------------------------------------------------------------
    s_med = int(np.argsort(span)[S // 2])
    bb, ee = b[s_med], e[s_med]
    t0 = np.nanmin(bb)
    streams = sorted({it["stream"] for it in items})
    main = plan.main.cuda_stream
    out.append("")
    out.append("(a) streams (the replay whose span is the median)")
    out.append(f"{'stream':>18s} {'kernels':>8s} {'busy ms':>9s} {'first':>8s} {'last':>8s} {'idle inside ms':>15s}  largest gaps (us: after -> before)")
    all_iv = []
    for st in streams:
        idx = [k for k in range(len(items)) if ok[k] and items[k]["stream"] == st]
        if not idx:
            continue
        iv = [(bb[k] - t0, ee[k] - t0) for k in idx]
        all_iv += iv
        busy = union_len(iv)
        first, last = min(a for a, _ in iv), max(c for _, c in iv)
        gaps = []
        order = sorted(idx, key=lambda k: bb[k])
        for p, q in zip(order, order[1:]):
            g = bb[q] - ee[p]
            if g > 0:
                gaps.append((g, p, q))
        gaps.sort(reverse=True)
        def released_by(q):   # the predecessor of q (schedule edges) that ended last: what the stream was waiting for
            cand = [r for r in preds[q] if ok[r]]
            if not cand:
                return "?"
            r = max(cand, key=lambda r: ee[r])
            return f"{short(items[r]['name'])}@{'main' if items[r]['stream'] == main else hex(items[r]['stream'])[-4:]}"
        gtxt = "; ".join(f"{g * 1e3:.0f}: {short(items[p]['name'])} -> {short(items[q]['name'])} [released by {released_by(q)}]" for g, p, q in gaps[:4])
        name = "main" if st == main else hex(st)[-6:]
        out.append(f"{name:>18s} {len(idx):8d} {busy:9.3f} {first:8.3f} {last:8.3f} {last - first - busy:15.3f}  {gtxt}")
    any_busy = union_len(all_iv)
    tot_busy = sum(c - a for a, c in all_iv)
    sp = float(span[s_med])
    out.append(f"some kernel running: {any_busy:.3f} of {sp:.3f} ms ({100 * any_busy / sp:.1f} %); NO kernel running: {sp - any_busy:.3f} ms "
               f"({100 * (1 - any_busy / sp):.1f} %); sum of kernel durations {tot_busy:.3f} ms = concurrency {tot_busy / any_busy:.2f}")
    res.update(no_kernel_ms=sp - any_busy, sum_kernel_ms=tot_busy)
    # same-stream hand-over gaps: distribution
    hg = []
    for st in streams:
        idx = sorted([k for k in range(len(items)) if ok[k] and items[k]["stream"] == st], key=lambda k: bb[k])
        for p, q in zip(idx, idx[1:]):
            # only pairs where q's other predecessors were done before p ended (a pure same-stream hand-over)
            if all((not ok[r]) or ee[r] <= ee[p] for r in preds[q]):
                hg.append(bb[q] - ee[p])
    hg = np.array(hg) * 1e3
    if len(hg):
        out.append(f"same-stream hand-over (end of a kernel -> begin of the next one that waited for nothing else): n={len(hg)}, median {np.median(hg):.1f} us, "
                   f"mean {hg.mean():.1f}, p90 {np.percentile(hg, 90):.1f}, sum {hg.sum() / 1e3:.3f} ms")
        res["handover_us_median"] = float(np.median(hg))
    # ---- per family: time in the step ---------------------------------------------------------------------------------------------
    fam = [family_of(it["name"], fams) if it["kind"] == 0 else "torch op (host-issued)" for it in items]
    out.append("")
    out.append("kernel families in the step (sum of in-step durations, median over the traced replays)")
    tab = {}
    for k in range(len(items)):
        if ok[k]:
            t = tab.setdefault(fam[k], [0, 0.0])
            t[0] += 1
            t[1] += dur[k]
    out.append(f"{'family':44s} {'launches':>8s} {'ms':>8s}")
    for f, (n, ms) in sorted(tab.items(), key=lambda x: -x[1][1]):
        out.append(f"{f:44s} {n:8d} {ms:8.3f}")
    out.append(f"{'total':44s} {sum(v[0] for v in tab.values()):8d} {sum(v[1] for v in tab.values()):8.3f}")
    res["family_ms"] = {f: round(v[1], 3) for f, v in tab.items()}
    # ---- (b) the chain that ends last --------------------------------------------------------------------------------------------
    # items without events (host-issued torch ops between plan segments, prof marks): their end is unknown -- they pass on the latest end
    # of their own predecessors, and what lies between that and the begin of their successor is booked on them
    eff = {}

    def eff_end(k):
        if k not in eff:
            eff[k] = ee[k] if ok[k] else max([eff_end(p) for p in preds[k]], default=t0)
        return eff[k]

    sys.setrecursionlimit(10000)
    endk = np.where(ok, ee, -np.inf)
    cur = int(np.argmax(endk))
    chain = []   # (item, begin - t0, duration, gap before it, cross-stream release)
    while True:
        ps = preds[cur]
        best = max(ps, key=eff_end) if ps else None
        rel = eff_end(best) if best is not None else t0
        if ok[cur]:
            chain.append((cur, bb[cur] - t0, ee[cur] - bb[cur], max(bb[cur] - rel, 0.0) if (best is None or ok[best]) else 0.0,
                          best is not None and items[best]["stream"] != items[cur]["stream"]))
        else:
            nxt = chain[-1][1] + t0 if chain else rel
            chain.append((cur, rel - t0, max(nxt - rel, 0.0), 0.0, False))
        if best is None:
            break
        cur = best
    chain.reverse()
    out.append("")
    out.append(f"(b) the dependency chain that ends last ({len(chain)} links; gap = begin - end of the predecessor that released it)")
    ch_fam, ch_gap_same, ch_gap_cross = {}, 0.0, 0.0
    for k, tb, d, g, cross in chain:
        ch_fam[fam[k]] = ch_fam.get(fam[k], 0.0) + d
        if cross:
            ch_gap_cross += g
        else:
            ch_gap_same += g
    tot_chain = sum(ch_fam.values()) + ch_gap_same + ch_gap_cross
    out.append(f"chain total {tot_chain:.3f} ms = kernels {sum(ch_fam.values()):.3f} + same-stream hand-overs {ch_gap_same:.3f} + cross-stream hand-overs {ch_gap_cross:.3f}")
    out.append(f"{'family on the chain':44s} {'ms':>8s} {'% of chain':>10s}")
    for f, ms in sorted(ch_fam.items(), key=lambda x: -x[1]):
        out.append(f"{f:44s} {ms:8.3f} {100 * ms / tot_chain:10.1f}")
    res["chain_family_ms"] = {f: round(v, 3) for f, v in ch_fam.items()}
    res["chain_gap_ms"] = round(ch_gap_same + ch_gap_cross, 3)
    out.append("")
    out.append(f"{'begin ms':>9s} {'dur us':>8s} {'gap us':>7s} {'stream':>7s}  kernel (grid)")
    info_cache = {}
    for k, tb, d, g, cross in chain:
        it = items[k]
        st = "main" if it["stream"] == main else hex(it["stream"])[-4:]
        grid = ""
        if it["kind"] == 0:
            grid = plan_grid(plan, it["node"], info_cache)
        out.append(f"{tb:9.3f} {d * 1e3:8.1f} {g * 1e3:7.1f}{'x' if cross else ' '} {st:>6s}  {short(it['name']) if it['kind'] == 0 else 'torch: ' + it['name']} {grid}")
    return res


def plan_grid(plan, node, cache):
    import ctypes
    from mgnet_amd import _C
    info = _C.PlanNodeInfo()
    if _C.lib().mgn_plan_node_info(plan.handle, node, ctypes.byref(info)) != 0:
        return ""
    return f"({info.grid[0]},{info.grid[1]},{info.grid[2]})x{info.block[0]}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "critical_path"))
    ap.add_argument("--steps", type=int, default=20, help="timed replays per what-if configuration")
    ap.add_argument("--traced", type=int, default=7)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--families", default="", help="extra what-if families name=regex,name=regex (matched before the built-in table)")
    ap.add_argument("--no-whatif", action="store_true")
    args = ap.parse_args()
    fams = [tuple(x.split("=", 1)) for x in args.families.split(",") if "=" in x] + FAMILIES
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    trainer, batch = build_trainer(args, dev)
    for _ in range(5):
        trainer.run_step(batch)
    plan = trainer.record_plan(batch)
    out = [f"critical_path.py: {args.batch} frames of {args.height}x{args.width}, {args.dtype}; plan: {json.dumps(plan.report)}"]
    t_plain = timed(trainer.replay_plan, args.steps, warm=5)
    t_eager = timed(lambda: trainer.run_step(batch), 10, warm=2)
    t_plain2 = timed(trainer.replay_plan, args.steps, warm=3)
    out.append(f"step time, un-traced plan replay: {t_plain:.3f} / {t_plain2:.3f} ms (two runs of {args.steps}); eager issue: {t_eager:.3f} ms")
    # ---- (a) + (b): traced replays ---------------------------------------------------------------------------------------------------
    plan.trace(True)
    t_traced = timed(trainer.replay_plan, args.steps, warm=3)
    out.append(f"step time with the per-dispatch events attached (back-to-back replays): {t_traced:.3f} ms (perturbation {t_traced - min(t_plain, t_plain2):+.3f} ms)")
    begins, ends = [], []
    for _ in range(args.traced):
        # two back-to-back replays, the SECOND one is read: its start overlaps the tail of its predecessor as in the timed loop
        trainer.replay_plan()
        trainer.replay_plan()
        torch.cuda.synchronize()
        bgn, end = plan.trace_read()
        begins.append(bgn)
        ends.append(end)
    plan.trace(False)
    res = analyse(plan, begins, ends, fams, out)
    res.update(step_ms=min(t_plain, t_plain2), step_eager_ms=t_eager, step_traced_ms=t_traced)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    np.savez_compressed(args.out + "_trace.npz", begins=np.array(begins), ends=np.array(ends),
                        names=np.array([it["name"] for it in plan.items]), nodes=np.array([it["node"] for it in plan.items]),
                        streams=np.array([it["stream"] for it in plan.items]), kinds=np.array([it["kind"] for it in plan.items]))
    # ---- (c) what-if -----------------------------------------------------------------------------------------------------------------
    if not args.no_whatif:
        items = plan.items
        fam = [family_of(it["name"], fams) if it["kind"] == 0 else None for it in items]
        opt_nodes = [it["node"] for it, f in zip(items, fam) if f == "optimizer"]
        plan.set_skip(opt_nodes, True)
        t_base = timed(trainer.replay_plan, args.steps, warm=3)
        out.append("")
        out.append("(c) what-if: step time with a family's launches left out (optimizer left out in every row so that the parameters stay finite;")
        out.append("    results of such a step are garbage, and kernels whose duration depends on their data -- reproj_march, ohem -- may run shorter)")
        out.append("    'twice' = the same launches issued twice instead: what the family costs, measured as an INCREASE on (mostly) intact data -- a step")
        out.append("    that computes on NaN / stale activations runs ~6 % faster (switching power -> clocks), so 'saves' over-states families whose")
        out.append("    absence poisons the activations downstream; * = the losses of the 'twice' step were not finite either")
        out.append(f"{'left out':44s} {'launches':>8s} {'in-step ms':>10s} {'step ms':>9s} {'saves ms':>9s} {'twice ms':>9s} {'costs ms':>9s}")
        out.append(f"{'optimizer only (base of the rows below)':44s} {len(opt_nodes):8d} {res['family_ms'].get('optimizer', 0.0):10.3f} {t_base:9.3f} {min(t_plain, t_plain2) - t_base:9.3f}")
        whatif = {"optimizer": {"step_ms": t_base, "saves_ms": min(t_plain, t_plain2) - t_base}}
        names = [f for f, _ in fams if f != "optimizer"] + ["other"]
        groups = [(f, [f]) for f in names]
        groups += [("ALL iabn", [f for f in names if f.startswith("iabn")]),
                   ("ALL conv fwd+dgrad", [f for f in names if f.startswith("conv ")] + ["stem fwd"]),
                   ("ALL wgrad", [f for f in names if "wgrad" in f]),
                   ("ALL stem (fwd, wgrad, pool)", [f for f in names if f.startswith("stem")]),
                   ("everything < 15 us in the step", None)]
        dur_item = np.nanmedian(np.array(ends) - np.array(begins), 0)
        bases = [t_base]
        # A what-if step computes on stale memory: NaN reaches the running statistics of the norm layers, which the conv epilogues use as
        # the shift of their statistics sums -- every later step then runs on NaN activations, and an all-NaN step is ~6 % FASTER (less
        # switching power, higher clocks: found when the base dropped 28.3 -> 26.6 ms behind the first conv what-if and stayed there).
        # So the model's parameters and buffers are restored after every row and the base's losses are checked to be finite.
        saved = {k: v.detach().clone() for k, v in trainer.model.state_dict().items()}

        def restore():
            with torch.no_grad():
                for k, v in trainer.model.state_dict().items():
                    v.copy_(saved[k])

        poisoned = 0
        for label, fs in groups:
            if fs is None:
                nodes = [it["node"] for it, f in zip(items, fam) if f is not None and f != "optimizer" and dur_item[it["node"]] < 0.015]
                instep = float(sum(dur_item[n] for n in nodes))
            else:
                nodes = [it["node"] for it, f in zip(items, fam) if f in fs]
                instep = sum(res["family_ms"].get(f, 0.0) for f in fs)
            if not nodes:
                continue
            plan.set_skip(nodes, True)
            try:
                t = timed(trainer.replay_plan, args.steps, warm=2)
            finally:
                plan.set_skip(nodes, False)
            restore()
            plan.set_skip(nodes, 2)      # the same family launched twice: its cost as an increase, on finite data
            try:
                t2 = timed(trainer.replay_plan, args.steps, warm=2)
                fin2 = all(bool(torch.isfinite(v).all()) for v in trainer._plan_losses.values())
            finally:
                plan.set_skip(nodes, False)
            restore()
            bases.append(timed(trainer.replay_plan, args.steps, warm=2))   # the base again after every row: the box drifts
            if not all(bool(torch.isfinite(v).all()) for v in trainer._plan_losses.values()):
                poisoned += 1
            base = 0.5 * (bases[-2] + bases[-1])
            whatif[label] = {"launches": len(nodes), "in_step_ms": instep, "step_ms": t, "base_ms": base, "saves_ms": base - t,
                             "twice_step_ms": t2, "twice_costs_ms": t2 - base, "twice_losses_finite": fin2}
            out.append(f"{label:44s} {len(nodes):8d} {instep:10.3f} {t:9.3f} {base - t:9.3f} {t2:9.3f} {t2 - base:9.3f}{' ' if fin2 else '*'}   (base {bases[-2]:.3f} / {bases[-1]:.3f})")
            print(out[-1], flush=True)
        out.append(f"base over the run: first {bases[0]:.3f}, last {bases[-1]:.3f}, min {min(bases):.3f}, max {max(bases):.3f} ms; base runs with non-finite losses: {poisoned}")
        res["whatif"] = whatif
        plan.set_skip(opt_nodes, False)
    txt = "\n".join(out)
    open(args.out + ".txt", "w").write(txt + "\n")
    json.dump(res, open(args.out + ".json", "w"), indent=1)
    print(txt)


if __name__ == "__main__":
    main()
