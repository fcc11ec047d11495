#!/usr/bin/env python
"""Summarise a rocprofv3 run (rocpd sqlite `*_results.db`, or `*_kernel_trace.csv`) into the
per-kernel table committed under profiles/ (calls, total, average, share), optionally restricted to
the steady-state frames (everything after the N-th composite_kernel launch).

    python tools/rocprof_summary.py gpurun_out/prof_x/x_results.db [--skip-frames 3] > profiles/rNN_x.txt
"""
import argparse
import csv
import sqlite3
import sys
from collections import defaultdict


def load(path):
    if path.endswith(".db"):
        c = sqlite3.connect(path)
        return [(n, s, e) for n, s, e in c.execute("select name, start, end from kernels order by start")]
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    return sorted(rows, key=lambda r: r[1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--skip-frames", type=int, default=0, help="drop everything up to the end of this many frames")
    ap.add_argument("--top", type=int, default=30)
    a = ap.parse_args()
    rows = load(a.path)
    frames = [e for n, s, e in rows if n.startswith("composite_kernel")]
    t_lo = frames[a.skip_frames - 1] if a.skip_frames and len(frames) >= a.skip_frames else 0
    nfr = len(frames) - (a.skip_frames if t_lo else 0)
    # the timed frames end with their composite kernel: what bench.py launches after the last one (its sigma>0
    # counting pass) is not part of a frame
    t_hi = frames[-1] if frames else (rows[-1][2] if rows else 0)
    agg = defaultdict(lambda: [0, 0])
    for n, s, e in rows:
        if s < t_lo or s >= t_hi:
            continue
        agg[n][0] += 1
        agg[n][1] += e - s
    tot = sum(v[1] for v in agg.values())
    print(f"# {a.path}: {nfr} frame(s) after skipping {a.skip_frames}; kernel time {tot/1e6:.2f} ms "
          f"({tot/1e6/max(nfr,1):.2f} ms/frame)")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'share':>7}  kernel")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[: a.top]:
        print(f"{c:7d} {t/1e6:10.3f} {t/1e3/c:10.2f} {100*t/tot:6.2f}%  {n[:110]}")
    # the dominant kernel's launches of the HEADLINE frame only: the run also holds the small launches of bench.py's extras and
    # spot checks, which drag the plain average down; a headline launch is one within 25 % of the longest
    if agg:
        top = max(agg.items(), key=lambda kv: kv[1][1])[0]
        d = [e - s for n, s, e in rows if n == top and t_lo <= s < t_hi]
        big = [x for x in d if x >= 0.75 * max(d)]
        print(f"# {top[:60]}: {len(big)} full-frame launches (within 25 % of the longest), average {sum(big)/len(big)/1e3:.1f} us, "
              f"min {min(big)/1e3:.1f}, max {max(big)/1e3:.1f}")


if __name__ == "__main__":
    sys.exit(main())
