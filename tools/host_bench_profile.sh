#!/bin/bash
# cProfile of bench.py's queueing loop (host cost per frame): tools/host_bench_profile.sh [bench args]  -> top functions by own time
mkdir -p gpurun_out
python -m cProfile -o /tmp/bench.prof bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 20 "$@" > /tmp/bench.out 2>/dev/null
tail -1 /tmp/bench.out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'host_pure', d['host_pure_ms_per_step'])"
cp /tmp/bench.prof gpurun_out/bench_${TAG:-x}.prof
python - <<'PY'
import pstats
st = pstats.Stats('/tmp/bench.prof')
st.sort_stats('tottime').print_stats(45)
PY
