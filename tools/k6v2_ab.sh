#!/bin/bash
# round 6: the 8-wave fused kernel against the 4-wave one on ONE box: image difference (whole headline frame), tile cycles, frame time
# usage (GPU box): bash tools/k6v2_ab.sh OUTDIR
out=$GRAFT_REPO_ROOT/$1
mkdir -p $out
cd $GRAFT_REPO_ROOT
for w in 4 8 4 8; do
  n=w$w; [ -e $out/$n.json ] && n=${n}b
  TH_FUSED_WAVES=$w TH_FUSED_DBG=1 TH_SAVE_IMAGE=$out/img_w$w.npy timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/$n.json 2> $out/$n.err
  python - "$n" "$out/$n.json" "$out/$n.err" <<'PY'
import json, sys
name, j, e = sys.argv[1:4]
try:
    d = json.loads(open(j).read().strip().split("\n")[-1])
    st = d["stage_ms_per_step"]
    dbg = [l for l in open(e) if "TH_FUSED_DBG" in l]
    tot = dbg[-1].split("total")[-1].strip() if dbg else "-"
    print(f"{name:6s} frame {d['ms_per_step']:.3f} ms  mlp {st['mlp']:.3f}  kernel {d.get('kernel_ms_per_step', 0):.3f} tile {tot}  dropin {d.get('dropin_ms_per_step', 0):.2f}")
    if dbg: print("   ", dbg[-1].split("barriers:")[-1].strip())
except Exception as ex:
    print(name, "FAILED", ex); print(open(e).read()[-3000:])
PY
done
python tools/cmp_imgs.py $out/img_w4.npy $out/img_w8.npy
