#!/bin/bash
# One gpurun call: default bench line (driver contract), per-phase cycle table of the fused kernel, PMC passes.
# usage: tools/gpu_baseline.sh gpurun_out/DIR
out=$GRAFT_REPO_ROOT/$1
mkdir -p $out
cd $GRAFT_REPO_ROOT
python bench.py > $out/bench.json 2> $out/bench.err
tail -c 1500 $out/bench.json
TH_FUSED_DBG=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $out/bench_dbg.json 2> $out/bench_dbg.err
grep TH_FUSED_DBG $out/bench_dbg.err
tools/pmc_mlp.sh $1/pmc_mlp
tools/pmc_hbm.sh $1/pmc_hbm
