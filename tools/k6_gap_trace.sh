#!/bin/bash
# tools/k6_gap_trace.sh OUT: rocprofv3 kernel trace of the timed frame loop; what runs between two launches of the fused kernel
# on the device timeline (tools/k6_gap_summary.py)
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/${1:-gpurun_out/k6gap}
mkdir -p $out
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $out -o trace -- python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5 > $out/bench.log 2>&1
ls $out | head
python tools/k6_gap_summary.py $out/trace_results.db
