#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for h in 9 1 5 4 8; do for nb in 148 64 256; do
  echo "scores hint $h nb $nb: $(HVR_DBG_KEYHINT=$h HVR_DBG_KEYNB=$nb timeout 100 python tools/key_bench.py --groups 4 2>&1 | grep 'grouped call')"
done; done
for h in 1 4 5 8; do
  echo "apply hint $h: $(HVR_DBG_KEYAHINT=$h timeout 100 python tools/key_bench.py --groups 4 2>&1 | grep 'grouped call')"
done
