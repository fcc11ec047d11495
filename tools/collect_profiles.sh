#!/bin/bash
# tools/collect_profiles.sh gpurun_out/DIR rNN_x : summaries of a tools/gpu_profiles.sh run -> profiles/ (tracked)
d=$1; tag=$2
cd "$(dirname "$0")/.."
tail -n 1 $d/bench.json > profiles/${tag}_bench.json
python tools/rocprof_summary.py $d/pmc_hbm/trace_results.db --skip-frames 3 > profiles/${tag}_kernel_stats.txt
python tools/pmc_hbm_summary.py $d/pmc_hbm > profiles/${tag}_pmc_hbm.txt
python tools/pmc_hbm_summary.py $d/pmc_hbm --json profiles/${tag}_pmc_hbm.txt > profiles/hbm_traffic.json
python tools/pmc_mlp_summary.py $d/pmc_mlp > profiles/${tag}_pmc_mlp.txt
ls -la profiles/${tag}_* profiles/hbm_traffic.json
