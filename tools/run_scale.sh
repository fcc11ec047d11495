#!/bin/bash
# 1 / 2 / 4 / 8 ranks on ONE node over RCCL (the driver's launch contract of bench.py), and for every N > 1 an A/B of
# the one collective that has never met hardware -- the 69 MB stem-latent broadcast of dist.StemExchange:
#   tools/run_scale.sh [OUTDIR] [STEPS] [WARMUP]
#     -> OUTDIR/scale_n1.json, OUTDIR/scale_n{2,4,8}_stem{0,1}.json (one JSON line each), the RCCL INFO log of every
#        job (OUTDIR/*.nccl.log: how many ranks initialised, which transport), OUTDIR/scale_summary.txt
# Needs as many visible GPUs as the largest rank count (RANKS="1 2 4 8" by default).  Every rank is one process on one
# GPU (LOCAL_RANK), the frame's pixel tiles are dealt to the ranks, TransHE of frame j runs on rank j mod N and its
# tokens are broadcast, the image is assembled with one all_gather (transhuman_amd/dist.py); value = rays of the frame /
# max-over-ranks time per step ("strong" scaling).  TH_STEM_EXCHANGE=0 is the default of bench.py (the safe variant).
# --check (default on; CHECK=0 switches it off): every job also saves the image of its last timed frame (rank 0, the gathered
# [R, 5] rgb | acc | depth; TH_SAVE_IMAGE) and the N-rank image is compared with the one-GPU image: max |diff| on rgb / acc must
# be <= 2e-6 (a ray shard equals the whole frame to fp32 rounding: the token blend runs on the matrix pipe per 32-sample tile,
# and a shard groups different samples into a tile).  The first hardware run then yields parity and the curve at once.
set -u
cd "$(dirname "$0")/.."
[ "${1:-}" = "--check" ] && shift
out=${1:-gpurun_out/scale}; steps=${2:-20}; warmup=${3:-3}
check=${CHECK:-1}
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
port=${MASTER_PORT:-29541}
: > "$out/scale_summary.txt"
summarise() {   # json-file  label  base-json
  python - "$1" "$2" "$3" <<'PY' | tee -a "$out/scale_summary.txt"
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    base = json.loads(open(sys.argv[3]).read().strip().split("\n")[-1])["value"]
    par = ""
    img, ref = sys.argv[1][:-5] + ".npy", sys.argv[3][:-5] + ".npy"
    import os
    if os.path.exists(img) and os.path.exists(ref) and img != ref:
        import numpy as np
        a, b = np.load(img), np.load(ref)
        dm = float(np.abs(a[:, :4].astype(np.float64) - b[:, :4]).max())
        par = f", image vs one GPU: max |rgb, acc| diff {dm:.2e} ({'OK' if dm <= 2e-6 else 'MISMATCH (bar 2e-6)'})"
    print(f"{sys.argv[2]}: {d['value'] / 1e6:.2f} M rays/s, {d['ms_per_step']:.3f} ms/step, x{d['value'] / base:.2f} over one GPU, "
          f"host {d.get('host_pure_ms_per_step', float('nan')):.2f} ms/step, stem_exchange={d['config'].get('stem_exchange')}" + par)
except Exception as e:          # noqa: BLE001
    print(f"{sys.argv[2]}: no JSON line ({e}); see the .err file")
PY
}
for n in ${RANKS:-1 2 4 8}; do
  avail=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
  # SCALE_FORCE=1: rehearsal on a one-GPU box (with TH_DIST_BACKEND=gloo TH_ONE_GPU=1 every rank uses cuda:0)
  if [ "$avail" -lt "$n" ] && [ "${SCALE_FORCE:-0}" != "1" ]; then echo "n=$n: only $avail GPU(s) visible, skipped" | tee -a "$out/scale_summary.txt"; continue; fi
  if [ "$n" = 1 ]; then
    TH_SAVE_IMAGE=$([ "$check" = 1 ] && echo "$out/scale_n1.npy") \
    python bench.py --gpus 1 --steps $steps --warmup $warmup --no-extras > "$out/scale_n1.json" 2> "$out/scale_n1.err"
    summarise "$out/scale_n1.json" "n=1" "$out/scale_n1.json"
    continue
  fi
  for stem in 0 1; do
    tag="n${n}_stem${stem}"
    TH_SAVE_IMAGE=$([ "$check" = 1 ] && echo "$out/scale_$tag.npy") TH_STEM_EXCHANGE=$stem NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT NCCL_DEBUG_FILE="$out/scale_$tag.nccl.%p.log" \
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
        bench.py --gpus $n --steps $steps --warmup $warmup > "$out/scale_$tag.json" 2> "$out/scale_$tag.err"
    port=$((port + 1))
    ranks=$(cat "$out"/scale_$tag.nccl.*.log 2>/dev/null | grep -c "Init COMPLETE" || true)
    cat "$out"/scale_$tag.nccl.*.log > "$out/scale_$tag.nccl.log" 2>/dev/null; rm -f "$out"/scale_$tag.nccl.*.log
    [ -f "$out/scale_n1.json" ] || { cp "$out/scale_$tag.json" "$out/scale_n1.json"; cp "$out/scale_$tag.npy" "$out/scale_n1.npy" 2>/dev/null; }
    summarise "$out/scale_$tag.json" "n=$n stem_exchange=$stem (RCCL communicators initialised: $ranks)" "$out/scale_n1.json"
  done
done
