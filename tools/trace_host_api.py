#!/usr/bin/env python
"""Host API calls of one frame of a rocprofv3 --hip-runtime-trace --kernel-trace run (rocpd sqlite), on the device timeline of
tools/trace_timeline.py: every call longer than --min us, and every graph launch / synchronisation, with its start offset from the
frame's first kernel.

    python tools/trace_host_api.py trace_results.db [--frame 6] [--min 15]
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--frame", type=int, default=6)
    ap.add_argument("--min", type=float, default=15.0)
    a = ap.parse_args()
    c = sqlite3.connect(a.path)
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    rows = [(n, s, e) for n, s, e in c.execute("select name, start, end from kernels order by start")]
    ends = [i for i, r in enumerate(rows) if r[0].startswith("composite_kernel")]
    lo, hi = ends[a.frame - 1] + 1, ends[a.frame] + 1
    t0, t1 = rows[lo][1], rows[hi - 1][2]
    src = "regions" if "regions" in names else None
    if src is None:
        print("no regions view; tables:", names)
        return
    cols = [r[1] for r in c.execute(f"pragma table_info({src})")]
    print("regions columns:", cols)
    q = f"select name, start, end from {src} where end >= ? and start <= ? order by start"
    for n, s, e in c.execute(q, (t0 - 300000, t1)):
        d = (e - s) / 1e3
        if d >= a.min or any(k in n for k in ("Graph", "Synchronize", "EventQuery")):
            print(f"{(s - t0) / 1e3:9.1f} us  +{d:8.1f}  {n}")


if __name__ == "__main__":
    main()
