#!/usr/bin/env python
"""Per-kernel durations of the last stand-alone ViT forward in a rocprofv3 trace (tools/vit_trace.sh):
    python tools/vit_trace_summary.py gpurun_out/DIR/vit_results.db"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = c.execute(f"select k.start,k.end,s.kernel_name from {kt} k join {ks} s on k.kernel_id=s.id order by k.start").fetchall()
# last forward = from the last add_kernel launch on
last = max(i for i, r in enumerate(rows) if 'add_kernel' in r[2])
rows = rows[last:]
agg = defaultdict(list)
for st, en, n in rows:
    agg[n[:48]].append((en - st) / 1e3)
for k, v in agg.items():
    print(f"{len(v):3d} x {sum(v)/len(v):7.1f} us = {sum(v):7.1f}  {k}")
print(f"span {(rows[-1][1]-rows[0][0])/1e3:.1f} us, kernel sum {sum(sum(v) for v in agg.values()):.1f} us, {len(rows)} launches")
