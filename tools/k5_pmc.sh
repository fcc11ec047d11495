#!/bin/bash
# PMC passes over K5 (pixgather_s256_kernel) on the frame's sample list and on the all-hit lists of tools/k5_locality.py:
#   tools/k5_pmc.sh OUTDIR   (on the GPU box) -> OUTDIR/pass_X_counter_collection.csv, OUTDIR/summary.txt
# dispatch order inside a pass: the cases of CASES, 6 launches each
out=$GRAFT_REPO_ROOT/$1
CASES=${CASES:-tile8,shuffled,window,one}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $out/avail.txt 2>&1
run() {
  name=$1; shift
  timeout ${PMC_TIMEOUT:-150} rocprofv3 --pmc "$@" --kernel-include-regex "pixgather_s256" --output-format csv -d $out -o pass_$name -- \
      python $GRAFT_REPO_ROOT/tools/k5_locality.py $CASES > $out/pass_$name.log 2>&1
}
run G GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES
run C1 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
run C2 TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run C3 TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum
run C4 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum
run D1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
run D2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum
run D3 TCC_READ_sum TCC_WRITE_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum
run F1 FETCH_SIZE TA_BUSY_avr TA_TA_BUSY_sum
run F2 WRITE_SIZE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run T1 TD_TD_BUSY_sum TD_TC_STALL_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum
python - $out $CASES <<'PY' | tee $out/summary.txt
import csv, glob, statistics, sys
from collections import defaultdict
d, cases = sys.argv[1], sys.argv[2].split(",")
for path in sorted(glob.glob(f"{d}/pass_*_counter_collection.csv")):
    rows = list(csv.DictReader(open(path)))
    if not rows: continue
    per = defaultdict(list)
    for r in rows: per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, v in sorted(per.items()):
        v.sort()
        n = len(v) // len(cases)
        line = f"{k:42s}"
        for i, c in enumerate(cases):
            vals = [x for _, x in v[i * n:(i + 1) * n]][1:]
            line += f" {c} {statistics.median(vals):16.0f}"
        print(line)
PY
