#!/bin/bash
# One gpurun call that produces everything the judged profiles come from (run on the GPU box):
#   tools/gpu_profiles.sh gpurun_out/DIR
#     DIR/bench.json            python bench.py (the driver's default line, with cpu_baseline and extras)
#     DIR/pmc_hbm/*             FETCH_SIZE / WRITE_SIZE passes + kernel trace   (tools/pmc_hbm.sh)
#     DIR/pmc_mlp/*             SQ / TCP / TCC passes over the fused MLP kernel (tools/pmc_mlp.sh)
# then, back in the repo:  tools/collect_profiles.sh gpurun_out/DIR rNN_x   -> profiles/rNN_x_*.txt|json
d=$1
out=$GRAFT_REPO_ROOT/$d
mkdir -p $out
cd $GRAFT_REPO_ROOT
python bench.py > $out/bench.json 2> $out/bench.err
tail -c 600 $out/bench.json
tools/pmc_hbm.sh $d/pmc_hbm > /dev/null
PMC_TIMEOUT=${PMC_TIMEOUT:-90} tools/pmc_mlp.sh $d/pmc_mlp > /dev/null
ls $out/pmc_mlp | tr '\n' ' '
