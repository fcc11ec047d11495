#!/bin/bash
# One box: the frame at N = 1 and the per-rank frame of emulated jobs of 2 / 4 / 8 ranks (no collectives) -> OUTDIR/*.json
#   tools/emu_scale.sh gpurun_out/DIR
cd $GRAFT_REPO_ROOT
out=$1; mkdir -p $out
for spec in "1 0 n1" "2 0 emu2" "4 1 emu4" "8 0 emu8_r0" "8 3 emu8_r3" "8 7 emu8_r7" "1 0 n1b"; do
  set -- $spec
  python bench.py --emulate-world $1 --emulate-rank $2 --no-cpu-baseline --no-extras > $out/$3.json 2> $out/$3.err
  python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().split(chr(10))[-1]); print(sys.argv[1], round(d['ms_per_step'],3), round(d['stage_ms_per_step']['mlp'],3), d.get('host_queue_ms_per_step'))" $out/$3.json
done
