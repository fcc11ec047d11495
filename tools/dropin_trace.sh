#!/bin/bash
# tools/dropin_trace.sh "ENV=VAL ..." ...: per environment a rocprofv3 kernel trace of tools/dropin_loop.py and the key marks of one
# render_fast frame's device timeline (tools/trace_timeline.py).  API=1: + HIP runtime API trace (tools/trace_host_api.py; slows the host)
mkdir -p /root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for e in "$@"; do
  i=$((i+1)); rm -rf /tmp/prof_d$i
  env $e rocprofv3 --kernel-trace ${API:+--hip-runtime-trace} --output-format rocpd -d /tmp/prof_d$i -- python /root/repo/tools/dropin_loop.py 6 > /tmp/prof_d$i.log 2>&1
  DB=$(find /tmp/prof_d$i -name "*.db" | head -1)
  echo "=== $e"; grep render_fast /tmp/prof_d$i.log
  python /root/repo/tools/trace_timeline.py $DB --frame 6 | cut -c1-110 | grep -v "bn_stats\|bn_apply\|conv_mfma\|gemm_h3<\|attn2\|map_box\|cmp_\|count_hits\|view_embed\|fold_color\|pack_linear\|at::native"
  [ -n "$API" ] && python /root/repo/tools/trace_host_api.py $DB --frame 6 --min 25 | cut -c1-160
done
