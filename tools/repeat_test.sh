#!/bin/bash
# failure statistics of one GPU test over N fresh processes:  tools/repeat_test.sh N <pytest node id> [ENV=value ...]
# (how the hipGraph memset-node problem was pinned down: profiles/r05_l_vit_graph_memset_node.txt)
N=${1:?runs}; T=${2:?pytest node id}; shift 2
f=0
for i in $(seq 1 "$N"); do
    env "$@" timeout 600 python -m pytest "$T" -x -q 2>&1 | tail -1 | grep -q failed && f=$((f+1))
done
echo "$T $* : $f / $N failed"
