#!/bin/bash
# tools/stem_ab.sh: the stem's statistics in the convolution epilogue (default) against TH_BN_STATS_PASS=1, interleaved:
# render_fast per call (30 calls each) and the stem graph alone
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for case in "epilogue|TH_X=0" "pass|TH_BN_STATS_PASS=1"; do
    name=${case%%|*}; envs=${case#*|}
    echo "== $name: $(env $envs python tools/dropin_loop.py 30 2>&1 | grep render_fast | tr '\n' ' ')"
  done
done
for case in "epilogue|TH_X=0" "pass|TH_BN_STATS_PASS=1"; do
  name=${case%%|*}; envs=${case#*|}
  echo "== stem alone, $name: $(env $envs python - <<'P'
import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device('cuda:0')
net = bench.build_net(dev)
enc = net.encoder
x = torch.rand(3, 3, 512, 512, device=dev)
with torch.no_grad():
    for _ in range(4): enc.trunk(x, graph=True)
    for mode in ('graph', 'eager'):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): enc.trunk(x, graph=(mode == 'graph'))
        torch.cuda.synchronize(); print(mode, 'us', round((time.perf_counter() - t) / 50 * 1e6, 1), end=' ')
P
)"
done
