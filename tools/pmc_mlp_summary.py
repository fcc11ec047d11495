#!/usr/bin/env python
"""Per-launch medians of the counters tools/pmc_mlp.sh collected over mlp_fused_kernel (full-size launches only:
the largest grid of the run), plus the derived ratios DESIGN.md quotes.

    python tools/pmc_mlp_summary.py gpurun_out/DIR > profiles/rNN_x_pmc_mlp.txt
"""
import csv
import glob
import statistics
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    vals = {}
    for path in sorted(glob.glob(f"{d}/pass_*_counter_collection.csv")):
        p = path.split("pass_")[1].split("_")[0]
        rows = list(csv.DictReader(open(path)))
        if not rows:
            continue
        gmax = max(int(r["Grid_Size"]) for r in rows)
        agg = defaultdict(list)
        for r in rows:
            if int(r["Grid_Size"]) == gmax:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        n = 0
        for k, v in sorted(agg.items()):
            vals[k] = statistics.median(v)
            n = len(v)
            print(f"{p}  {k:38s} {vals[k]:18.0f}")
        wg = int(rows[0].get("Workgroup_Size", 256) or 256)
        vals["_waves_per_simd"] = max(1.0, wg / 256.0)
        print(f"#  pass {p}: {n} launches of {gmax // wg} tiles ({wg}-thread workgroups), VGPRs {rows[0]['VGPR_Count']}+{rows[0]['Accum_VGPR_Count']}, "
              f"scratch {rows[0]['Scratch_Size']} B/lane, LDS {rows[0]['LDS_Block_Size']}")
    g = vals.get
    if g("SQ_WAVE_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
        # SQ_* count quad-cycles (4 clocks) except SQ_VALU_MFMA_BUSY_CYCLES (clocks, summed over the SIMDs)
        # time base of the matrix pipe = SIMD cycles: wave cycles / waves per SIMD (one wave per SIMD for 256-thread workgroups, two for 512)
        wps = g("_waves_per_simd", 1.0)
        wave_clk = 4.0 * g("SQ_WAVE_CYCLES") / wps
        print(f"# derived: MFMA busy = {g('SQ_VALU_MFMA_BUSY_CYCLES'):.3e} / (4 * {g('SQ_WAVE_CYCLES'):.3e} / {wps:.0f} waves per SIMD) = "
              f"{100.0 * g('SQ_VALU_MFMA_BUSY_CYCLES') / wave_clk:.1f} % of SIMD cycles; the percentages below are of WAVE cycles")
        if g("SQ_WAIT_INST_ANY"):
            print(f"#          waiting on any counter {100.0 * g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.1f} %, "
                  f"on LDS {100.0 * g('SQ_WAIT_INST_LDS', 0) / g('SQ_WAVE_CYCLES'):.1f} %, "
                  f"VALU issue {100.0 * g('SQ_ACTIVE_INST_VALU', 0) / g('SQ_WAVE_CYCLES'):.1f} %, "
                  f"LDS issue {100.0 * g('SQ_ACTIVE_INST_LDS', 0) / g('SQ_WAVE_CYCLES'):.1f} %")
    if g("SQ_LDS_BANK_CONFLICT") and g("SQ_LDS_IDX_ACTIVE"):
        print(f"#          LDS bank-conflict cycles / LDS active cycles = "
              f"{100.0 * g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'):.1f} %")
    if g("TCC_HIT_sum") and g("TCC_MISS_sum"):
        print(f"#          L2 hit rate = {100.0 * g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.1f} % of "
              f"{g('TCC_REQ_sum', 0):.3e} requests per launch")
    if g("TCP_TCC_READ_REQ_sum") and g("TCP_TCC_READ_REQ_LATENCY_sum"):
        print(f"#          L1 -> L2 read latency = {g('TCP_TCC_READ_REQ_LATENCY_sum') / g('TCP_TCC_READ_REQ_sum'):.0f} cycles "
              f"(average over {g('TCP_TCC_READ_REQ_sum'):.3e} requests)")
    if g("FETCH_SIZE") and g("WRITE_SIZE"):
        print(f"#          HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE = {(2 * g('FETCH_SIZE') + g('WRITE_SIZE')) * 1024 / 1e9:.3f} GB "
              f"(KB counters; gfx950 correction of MI355X_MICROARCH.md)")
    if g("SQ_INSTS_MFMA"):
        print(f"#          instructions per launch: MFMA {g('SQ_INSTS_MFMA'):.3e}, VALU {g('SQ_INSTS_VALU', 0):.3e}, "
              f"SALU {g('SQ_INSTS_SALU', 0):.3e}")


if __name__ == "__main__":
    main()
