#!/usr/bin/env python
"""Join the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_hbm.sh into a per-kernel HBM table.

    python tools/pmc_hbm_summary.py gpurun_out/prof_x > profiles/rNN_x_pmc_hbm.txt

Units: KB per launch.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes
of wide coalesced reads, WRITE_SIZE is exact -> hbm = 2 * FETCH + WRITE.
"""
import csv
import sys
from collections import defaultdict


def load(path, name):
    agg = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    d = sys.argv[1]
    as_json = "--json" in sys.argv
    f = load(f"{d}/pmc_FETCH_SIZE_counter_collection.csv", "FETCH_SIZE")
    w = load(f"{d}/pmc_WRITE_SIZE_counter_collection.csv", "WRITE_SIZE")
    if not as_json:            # (the --json form prints the JSON object only: bench.py json.load()s the file)
        print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1")
        print("# KB per launch = MAX over the launches of a kernel (one launch covers the frame's valid samples); hbm = 2*FETCH + WRITE")
        print(f"{'kernel':62s} {'launches':>8s} {'FETCH_KB':>12s} {'WRITE_KB':>12s} {'HBM_GB(2F+W)':>13s}")
    rows = []
    for k in f:
        fk = max(f[k])
        wk = max(w.get(k, [0.0]))
        rows.append((2 * fk + wk, k, len(f[k]), fk, wk))
    if as_json:
        # profiles/hbm_traffic.json (read by bench.py): bytes per full launch of the three per-sample kernels
        import json

        def of(sub):
            m = [r for r in rows if sub in r[1]]
            return max(m)[0] * 1024 if m else None
        src = sys.argv[sys.argv.index("--json") + 1] if len(sys.argv) > sys.argv.index("--json") + 1 else d
        # samples of the (one) launch per kernel and frame: the valid-sample count of the profiled frame, from the JSON
        # line bench.py printed under the profiler (ABI 6: one launch of K4 / K5 / K6 covers the whole frame)
        ch, V = 524288, 3
        try:
            line = [l for l in open(f"{d}/pmc_FETCH_SIZE.log").read().splitlines() if l.startswith("{")][-1]
            ch = int(json.loads(line)["config"]["valid_samples_rank0"])
        except (OSError, IndexError, KeyError, ValueError):
            pass
        tex = of("pixtex_kernel") is not None            # TH_ROWS_TEX: K5t instead of K5, no pixel-feature rows through HBM
        hw = 512 * 512
        try:
            hw = int(json.loads(line)["config"]["rays"])
        except (NameError, KeyError, ValueError):
            pass
        if tex:
            alg = ch * (256 + 64 + V * 32) + ch // 32 * 1024 + 2 * V * hw * 1024
            note = ("positional encoding 256 B, neighbour record 64 B, three 32-byte texel records per sample + a 512 B token header "
                    "and a 512 B texel list per 32-sample tile + every texel of the two folded maps once (2 x V x H x W x 1 KiB: an "
                    "upper bound, the maps are cropped to the hull)")
        else:
            alg = ch * (V * 1088 + 256 + 64) + ch // 32 * 512
            note = ("pixel-feature rows once (3 x 1088 B), positional encoding 256 B, neighbour record 64 B per sample + a 512 B "
                    "header per 32-sample tile")
        print(json.dumps({"source": src, "launch_samples": ch,
                          "mlp_fused_bytes_per_launch": of("mlp_fused"),
                          "gather_kernel": "pixtex_kernel" if tex else "pixgather_s256_kernel",
                          "pixgather_bytes_per_launch": of("pixtex_kernel") or of("pixgather_s256") or of("pixgather_kernel<true>"),
                          "dparf_bytes_per_launch": of("dparf_kernel"),
                          "mlp_fused_algorithmic_bytes_per_launch": alg,
                          "algorithmic_note": note}, indent=1))
        return
    for hbm, k, n, fk, wk in sorted(rows, reverse=True)[:16]:
        print(f"{k[:62]:62s} {n:8d} {fk:12.0f} {wk:12.0f} {hbm * 1024 / 1e9:13.3f}")


if __name__ == "__main__":
    main()
