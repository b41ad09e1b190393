#!/bin/bash
# per-kernel time of the relation passes vs K: D sweep (scores K = D) and Mk sweep (apply K = Mk)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
run() {
  rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats -d /tmp/ks -o r -- python tools/rel_bench.py --iters 5 $1 > /tmp/ks.log 2>&1
  DB=$(find /tmp/ks -name "*.db" | head -1); echo "=== $1  $(grep ^relation /tmp/ks.log)"
  python tools/rocpd_stats.py $DB | grep "tile_kernel\|transpose" | cut -c1-60,112-170
}
for d in 256 512 1024 2048; do run "--d $d"; done
for mk in 1152 2304 4608 9216; do run "--mk $mk"; done
