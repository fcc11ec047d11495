#!/usr/bin/env python
"""Between two full-frame launches of the fused kernel (steady state of the frame loop): the idle time of the device, and the
kernels that run alone there (nothing else of the frame can hide them).

    python tools/k6_gap_summary.py gpurun_out/k6gap/trace_results.db
"""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
rows = [(n, s, e) for n, s, e in c.execute("select name, start, end from kernels order by start")]
k6 = [(s, e) for n, s, e in rows if "mlp_fused" in n]
longest = max(e - s for s, e in k6)
k6 = [(s, e) for s, e in k6 if e - s > 0.75 * longest]
k6 = k6[8:-2]                                   # steady state
print(f"{len(k6)} full-frame launches, average {sum(e - s for s, e in k6) / len(k6) / 1e3:.1f} us; "
      f"period {(k6[-1][0] - k6[0][0]) / (len(k6) - 1) / 1e3:.1f} us")
gap_tot, alone = 0.0, defaultdict(float)
idle_tot = 0.0
for (s0, e0), (s1, e1) in zip(k6, k6[1:]):
    gap_tot += s1 - e0
    inside = sorted((max(s, e0), min(e, s1), n) for n, s, e in rows if e > e0 and s < s1 and "mlp_fused" not in n)
    t = e0
    for s, e, n in inside:
        if s > t:
            idle_tot += s - t
        if e > t:
            alone[n.split("(")[0][:60]] += e - max(s, t)
            t = e
    if s1 > t:
        idle_tot += s1 - t
n = len(k6) - 1
print(f"between launches: {gap_tot / n / 1e3:.1f} us per frame, of which the device is idle {idle_tot / n / 1e3:.1f} us")
for name, t in sorted(alone.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {t / n / 1e3:8.1f} us  {name}")
