#!/bin/bash
# HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE in separate rocprofv3 runs, every kernel) + a kernel trace.
# usage (on the GPU box, through gpurun): tools/pmc_hbm.sh gpurun_out/DIR
out=$GRAFT_REPO_ROOT/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $c --output-format csv -d $out -o pmc_$c -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $out/pmc_$c.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats -d $out -o trace -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-extras > $out/trace_bench.log 2>&1
ls $out
