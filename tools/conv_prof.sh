#!/bin/bash
# per-kernel average duration of the conv kernels for a library variant: tools/conv_prof.sh [lib.so]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cvp; TH_LIB_PATH=$1 timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/cvp -o trace -- python $GRAFT_REPO_ROOT/tools/conv_time.py > /tmp/cvp.txt 2>&1
f=$(find /tmp/cvp -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $f --top 40 | grep -i "conv.*_mfma\|maxpool" | cut -c1-100
