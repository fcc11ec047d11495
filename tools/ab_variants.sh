#!/bin/bash
# tools/ab_variants.sh NAME...: per variant library transhuman_amd/_variants/libNAME.so (tools/build_variant.sh) the fused kernel's
# cycles between barriers (TH_FUSED_DBG) and the frame time of a short bench run
for n in "$@"; do
  L=transhuman_amd/_variants/lib$n.so
  [ "$n" = base ] && L=transhuman_amd/libtranshuman_hip.so
  echo "== $n"
  TH_LIB_PATH=$PWD/$L TH_FUSED_DBG=1 python bench.py --steps 12 --warmup 3 --no-extras --no-cpu-baseline 2>&1 | grep -a -E "TH_FUSED_DBG|ms_per_step" | sed -E 's/.*(average cycles between barriers:.*)/\1/; s/.*"ms_per_step": ([0-9.]+).*"ms_per_step_median": ([0-9.]+).*/ms_per_step \1 median \2/'
done
