#!/bin/bash
# round 6: tiles per workgroup of the 8-wave kernel (TH_FUSED_TPW) against the 4-wave kernel, one box, alternating
out=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $out
cd $GRAFT_REPO_ROOT
for case in "$@"; do
  name=${case%%|*}; envs=${case#*|}
  env $envs timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $out/$name.json 2> $out/$name.err
  python - "$name" "$out/$name.json" <<'PY'
import json, sys
name, j = sys.argv[1:3]
try:
    d = json.loads(open(j).read().strip().split("\n")[-1])
    st = d["stage_ms_per_step"]
    print(f"{name:10s} frame {d['ms_per_step']:.3f} ms (min {d['ms_per_step_min']:.2f} med {d['ms_per_step_median']:.2f})  mlp {st['mlp']:.3f}  dropin {d.get('dropin_ms_per_step', 0):.2f}  clock {d['shader_clock_GHz']['median']:.3f}")
except Exception as ex:
    print(name, "FAILED", ex)
PY
done
