#!/bin/bash
# per-kernel times of the full relation stage; usage: rel_prof.sh "ENV=.. ENV=.. -- args" ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  envs=${cfg%%--*}; args=${cfg#*--}
  rm -rf /tmp/rp; env $envs rocprofv3 --kernel-trace --stats -d /tmp/rp -o r -- python tools/rel_bench.py --iters 5 $args > /tmp/rp.log 2>&1
  DB=$(find /tmp/rp -name "*.db" | head -1); echo "=== $cfg  $(grep ^relation /tmp/rp.log)"
  [ -n "$DB" ] && python tools/rocpd_stats.py $DB | grep "tile_kernel\|transpose\|relation" | cut -c1-60,112-170 || tail -3 /tmp/rp.log
done
