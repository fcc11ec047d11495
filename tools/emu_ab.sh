#!/bin/bash
# tools/emu_ab.sh "ENV=VAL ..." ["ENV=VAL ..." ...]: per environment the N = 1 frame and the emulated rank 3 of 8 (bench.py), one box
one() { python bench.py "$@" --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); s=d['stage_ms_per_step']; print(round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median'],3), 'mlp', round(s['mlp'],3), 'fold', round(s.get('fold',0),3), 'host', round(d['host_pure_ms_per_step'],3), 'dropin', d.get('dropin_ms_per_step'))"; }
for e in "$@"; do
  echo "== $e"
  echo -n "  n1    : "; env $e bash -c "$(declare -f one); one"
  echo -n "  emu8r3: "; env $e bash -c "$(declare -f one); one --emulate-world 8 --emulate-rank 3"
done
