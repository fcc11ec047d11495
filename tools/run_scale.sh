#!/bin/bash
# 1 / 2 / 4 / 8 ranks on ONE node over RCCL, one JSON line each (the driver's launch contract of bench.py):
#   tools/run_scale.sh [OUTDIR] [STEPS] [WARMUP]      -> OUTDIR/scale_n{1,2,4,8}.json + OUTDIR/scale_summary.txt
# Needs as many visible GPUs as the largest rank count it is asked for (RANKS="1 2 4 8" by default).  Every rank is one
# process on one GPU (LOCAL_RANK), the frame's pixel tiles are dealt to the ranks, TransHE of frame j runs on rank j mod N and
# its tokens are broadcast, the image is assembled with one all_gather (transhuman_amd/dist.py); value = rays of the frame /
# max-over-ranks time per step ("strong" scaling).
set -u
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/scale}; steps=${2:-20}; warmup=${3:-3}
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
port=${MASTER_PORT:-29541}
: > "$out/scale_summary.txt"
for n in ${RANKS:-1 2 4 8}; do
  avail=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
  if [ "$avail" -lt "$n" ]; then echo "n=$n: only $avail GPU(s) visible, skipped" | tee -a "$out/scale_summary.txt"; continue; fi
  if [ "$n" = 1 ]; then
    python bench.py --gpus 1 --steps $steps --warmup $warmup --no-extras > "$out/scale_n1.json" 2> "$out/scale_n1.err"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
        bench.py --gpus $n --steps $steps --warmup $warmup > "$out/scale_n$n.json" 2> "$out/scale_n$n.err"
    port=$((port + 1))
  fi
  python - "$out/scale_n$n.json" "$n" "$out/scale_n1.json" <<'PY' | tee -a "$out/scale_summary.txt"
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    base = json.loads(open(sys.argv[3]).read().strip().split("\n")[-1])["value"] if sys.argv[2] != "1" else d["value"]
    print(f"n={sys.argv[2]}: {d['value'] / 1e6:.2f} M rays/s, {d['ms_per_step']:.3f} ms/step, x{d['value'] / base:.2f} over one GPU")
except Exception as e:          # noqa: BLE001
    print(f"n={sys.argv[2]}: no JSON line ({e}); see the .err file")
PY
done
