#!/bin/bash
# kernel trace of the stand-alone TransHE forward (tools/vit_time.py): per-kernel durations and gaps
# usage (GPU box): tools/vit_trace.sh gpurun_out/DIR
out=$GRAFT_REPO_ROOT/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/tools/vit_time.py 500 1500 > $out/vit_time.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $out -o vit -- python $GRAFT_REPO_ROOT/tools/vit_time.py 500 > $out/vit_trace.log 2>&1
ls $out
