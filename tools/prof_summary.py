#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) into the per-kernel table kept under profiles/."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
        "group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = [f"{'kernel':110s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}"]
    for n, k, s, a, mn, mx in rows:
        lines.append(f"{n[:110]:110s} {k:6d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*s/tot:6.2f}")
    # GPU idle time between consecutive kernels over the last 60 % of the trace (steady state: no model set-up)
    ks = c.execute("select start, end from kernels order by start").fetchall()
    if len(ks) > 100:
        ks = ks[int(len(ks) * 0.4):]
        span = ks[-1][1] - ks[0][0]
        busy, last_end, gaps = 0, ks[0][0], 0
        for st, en in ks:
            if st > last_end:
                gaps += st - last_end
            busy += en - st
            last_end = max(last_end, en)
        lines.append(f"steady-state window: span {span/1e6:.2f} ms, kernel time {busy/1e6:.2f} ms, idle between kernels {gaps/1e6:.2f} ms "
                     f"({100*gaps/span:.1f} %), {len(ks)} launches, mean gap {gaps/len(ks)/1e3:.2f} us")
    # overlapped execution (side streams): union of the kernel intervals vs their sum over the same steady-state window
    if len(ks) > 100:
        ev = sorted(ks)
        union, cur_s, cur_e = 0, ev[0][0], ev[0][1]
        for st, en in ev[1:]:
            if st > cur_e:
                union += cur_e - cur_s
                cur_s, cur_e = st, en
            else:
                cur_e = max(cur_e, en)
        union += cur_e - cur_s
        tot = sum(en - st for st, en in ev)
        span = max(e for _, e in ev) - ev[0][0]
        lines.append(f"concurrency: span {span/1e6:.2f} ms, at least one kernel running {union/1e6:.2f} ms ({100*union/span:.1f} %), "
                     f"sum of kernel times {tot/1e6:.2f} ms (mean concurrency {tot/union:.2f})")
    # the largest idle gaps of the steady-state window and the kernels on either side of each (what was the GPU waiting for?)
    named = c.execute("select start, end, name from kernels order by start").fetchall()
    if len(named) > 100:
        named = named[int(len(named) * 0.4):]
        gaps, last_end, last_name = [], named[0][1], named[0][2]
        for st, en, nm in named[1:]:
            if st > last_end:
                gaps.append((st - last_end, last_name, nm))
            if en >= last_end:
                last_end, last_name = en, nm
        # idle time of the steady-state window by the kernel that ENDS the gap (what the chip was waiting to start) and by the one before it
        by_next, by_prev = {}, {}
        for g, a, b in gaps:
            if g < 200e3:   # (not the gaps between phases of the script)
                by_next[b[:70]] = by_next.get(b[:70], [0, 0]); by_next[b[:70]][0] += g; by_next[b[:70]][1] += 1
                by_prev[a[:70]] = by_prev.get(a[:70], [0, 0]); by_prev[a[:70]][0] += g; by_prev[a[:70]][1] += 1
        lines.append("idle time by the kernel that starts after the gap (ms total, gaps, mean us):")
        for k, (t, n) in sorted(by_next.items(), key=lambda kv: -kv[1][0])[:14]:
            lines.append(f"  {t/1e6:8.2f} {n:6d} {t/n/1e3:7.1f}  {k}")
        lines.append("idle time by the kernel that ran before the gap:")
        for k, (t, n) in sorted(by_prev.items(), key=lambda kv: -kv[1][0])[:14]:
            lines.append(f"  {t/1e6:8.2f} {n:6d} {t/n/1e3:7.1f}  {k}")
        gaps.sort(reverse=True)
        lines.append("largest idle gaps (us): after kernel -> before kernel")
        for g, a, b in gaps[:24]:
            lines.append(f"  {g/1e3:9.1f}  {a[:60]:60s} -> {b[:60]}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
