#!/usr/bin/env python
"""Two-stream timeline of one steady-state frame of a rocprofv3 kernel trace (rocpd sqlite): every kernel between
two consecutive composite_kernel launches with start offset, duration and queue, runs of the same kernel merged.

    python tools/trace_timeline.py gpurun_out/prof_x/trace_results.db [--frame 5]
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--frame", type=int, default=5)
    a = ap.parse_args()
    c = sqlite3.connect(a.path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = [(n, s, e, qq) for n, s, e, qq in c.execute(f"select name, start, end, {q} from kernels order by start")]
    ends = [i for i, r in enumerate(rows) if r[0].startswith("composite_kernel")]
    lo, hi = ends[a.frame - 1] + 1, ends[a.frame] + 1
    t0 = rows[lo][1]
    print(f"frame {a.frame}: {(rows[hi - 1][2] - rows[lo - 1][2]) / 1e6:.3f} ms between composite ends; columns: {cols}")
    cur = None
    for n, s, e, qq in rows[lo:hi]:
        key = (n[:40], qq)
        if cur and cur[0] == key:
            cur[2] = e; cur[3] += 1; cur[4] += e - s
        else:
            if cur:
                print(f"{(cur[1] - t0) / 1e3:9.1f} us  +{(cur[2] - cur[1]) / 1e3:8.1f}  busy {cur[4] / 1e3:8.1f}  x{cur[3]:<3d} q{cur[0][1]}  {cur[0][0]}")
            cur = [key, s, e, 1, e - s]
    if cur:
        print(f"{(cur[1] - t0) / 1e3:9.1f} us  +{(cur[2] - cur[1]) / 1e3:8.1f}  busy {cur[4] / 1e3:8.1f}  x{cur[3]:<3d} q{cur[0][1]}  {cur[0][0]}")


if __name__ == "__main__":
    main()
