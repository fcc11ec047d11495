#!/bin/bash
# PMC passes over the fused MLP kernel (run on the GPU box through gpurun; one rocprofv3 run per counter set).
# usage: tools/pmc_mlp.sh OUTDIR   -> OUTDIR/pass_X_counter_collection.csv
out=$GRAFT_REPO_ROOT/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  # PMC_BENCH_ARGS: e.g. --no-pipeline (one stream: the TCP / TCC passes of rocprofv3 abort with 'incomplete dispatches' on the
  # two-stream frame pipeline); PMC_TIMEOUT: seconds per pass
  timeout ${PMC_TIMEOUT:-200} rocprofv3 --pmc "$@" --kernel-include-regex "${KREGEX:-mlp_fused}" --output-format csv -d $out -o pass_$name -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras $PMC_BENCH_ARGS > $out/pass_$name.log 2>&1
}
tcc_passes() {   # TCP / TCC / traffic: at most 3-4 counters of a block per pass (more: 'exceeds the capabilities of the hardware')
  run C1 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
  run C2 TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
  run D1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
  run D2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum
  run F1 FETCH_SIZE TA_BUSY_avr
  run F2 WRITE_SIZE TCP_TCC_WRITE_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum
}
if [ "$2" = "TCC" ]; then tcc_passes; ls $out; exit 0; fi
if [ -n "$2" ]; then shift; name=$1; shift; run $name "$@"; ls $out; exit 0; fi
run A SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
run B SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run E SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_TC_STALL SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU
tcc_passes
ls $out
