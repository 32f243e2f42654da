#!/usr/bin/env python3
"""Instruction budget of a loop of a kernel from hipcc's -S output: tools/isa_budget.py <file.s> <first line> <last line> [title].
Classes follow the issue cost measured by tools/probe/valu_probe.hip on MI355X (cycles per wave-instruction per SIMD)."""
import re, sys, collections
FAST = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mov_b32", "v_mov_b64", "v_and_b32", "v_or_b32", "v_xor_b32")
def cls(m, line):
    if m.startswith(("buffer_load", "global_load", "scratch_load")): return "VMEM load"
    if m.startswith(("buffer_store", "global_store", "scratch_store")): return "VMEM store"
    if m.startswith("ds_"): return "LDS"
    if m.startswith("s_waitcnt"): return "s_waitcnt"
    if m.startswith("s_barrier"): return "s_barrier"
    if m.startswith("s_"): return "SALU / branch"
    if "dpp" in line: return "VALU dpp (wave shift)"
    if m.startswith(("v_rcp", "v_exp", "v_log", "v_sqrt", "v_rsq")): return "VALU transcendental"
    if m.startswith("v_pk_"): return "VALU packed f32"
    if m.startswith("v_cndmask"): return "VALU select"
    if m.startswith("v_cmp"): return "VALU compare"
    if m.startswith(FAST): return "VALU full rate (add/mul/fma/mov)"
    if m.startswith("v_"): return "VALU other (min/max/med3/cvt/floor/int)"
    return "other"
f, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cnt, det = collections.Counter(), collections.defaultdict(collections.Counter)
for ln in open(f).read().split("\n")[a - 1:b]:
    m = re.match(r"^\s+([a-z][a-z0-9_]+)", ln)
    if m:
        c = cls(m.group(1), ln); cnt[c] += 1; det[c][m.group(1)] += 1
print(sys.argv[4] if len(sys.argv) > 4 else f"{f}:{a}-{b}")
tot = sum(v for k, v in cnt.items() if k.startswith("VALU"))
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(f"  {k:42s} {v:5d}   " + ", ".join(f"{n} x{c}" for n, c in det[k].most_common(6)))
print(f"  VALU total {tot}")
