#!/bin/bash
# tools/dropin_quick.sh "name|ENV=.." ...: render_fast per call (tools/dropin_loop.py) per environment
cd $GRAFT_REPO_ROOT
for case in "$@"; do
  name=${case%%|*}; envs=${case#*|}
  echo "== $name: $(env $envs python tools/dropin_loop.py 6 2>&1 | grep render_fast | tr '\n' ' ')"
done
