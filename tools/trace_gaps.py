#!/usr/bin/env python
"""Idle gaps on the GPU timeline of a rocprofv3 kernel trace (rocpd sqlite): for the steady-state frames
(after --skip-frames composite_kernel launches) print wall time, summed kernel time and the largest gaps
with the kernels on either side.

    python tools/trace_gaps.py gpurun_out/prof_x/trace_results.db --skip-frames 3
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--skip-frames", type=int, default=3)
    ap.add_argument("--top", type=int, default=14)
    a = ap.parse_args()
    c = sqlite3.connect(a.path)
    rows = [(n, s, e) for n, s, e in c.execute("select name, start, end from kernels order by start")]
    ends = [i for i, r in enumerate(rows) if r[0].startswith("composite_kernel")]
    lo = ends[a.skip_frames - 1] + 1 if a.skip_frames > 0 else 0
    hi = ends[-1] + 1
    rows = rows[lo:hi]
    nfr = len(ends) - a.skip_frames
    wall = (rows[-1][2] - rows[0][1]) / 1e6
    busy = sum(e - s for _, s, e in rows) / 1e6
    print(f"{nfr} frames: wall {wall:.2f} ms ({wall / nfr:.2f}/frame), kernels {busy:.2f} ms ({busy / nfr:.2f}/frame), "
          f"idle {wall - busy:.2f} ms ({(wall - busy) / nfr:.2f}/frame), {len(rows) / nfr:.0f} launches/frame")
    gaps = []
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        gaps.append(((s1 - e0) / 1e3, n0[:48], n1[:48]))
    gaps.sort(reverse=True)
    for g, n0, n1 in gaps[:a.top]:
        print(f"{g:9.1f} us  after {n0:48s} before {n1}")
    import collections
    hist = collections.Counter()
    for g, _, _ in gaps:
        hist["<2us" if g < 2 else "<5us" if g < 5 else "<10us" if g < 10 else "<50us" if g < 50 else ">=50us"] += g
    print({k: round(v / nfr) for k, v in hist.items()}, "us/frame by gap size")


if __name__ == "__main__":
    main()
