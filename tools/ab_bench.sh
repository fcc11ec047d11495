#!/bin/bash
# A/B of library variants / env switches on ONE box: tools/ab_bench.sh OUTDIR "name|ENV=.. ENV=.." ...
# each case: bench.py --steps 8 --warmup 3 with TH_FUSED_DBG=1 -> one summary line (frame ms, MLP ms, gather, dparf, tile cycles)
out=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $out
cd $GRAFT_REPO_ROOT
for case in "$@"; do
  name=${case%%|*}; envs=${case#*|}
  env $envs TH_FUSED_DBG=1 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $out/$name.json 2> $out/$name.err
  python - "$name" "$out/$name.json" "$out/$name.err" <<'PY'
import json, sys
name, j, e = sys.argv[1:4]
try:
    d = json.loads(open(j).read().strip().split("\n")[-1])
    st = d["stage_ms_per_step"]
    dbg = [l for l in open(e) if "TH_FUSED_DBG" in l]
    tot = dbg[-1].split("total")[-1].strip() if dbg else "-"
    print(f"{name:12s} frame {d['ms_per_step']:.3f} ms  mlp {st['mlp']:.3f}  gather {st['gather']:.3f}  dparf {st['dparf']:.3f}  hull {st['hull']:.2f} vit {st['vit']:.2f}  frac {d['roofline']['frac']:.4f}  tile {tot}  rf {d.get('render_fast_ms_per_step', 0):.2f}")
    if dbg: print("   ", dbg[-1].split("barriers:")[-1].strip())
except Exception as ex:
    print(name, "FAILED", ex); print(open(e).read()[-1500:])
PY
done
