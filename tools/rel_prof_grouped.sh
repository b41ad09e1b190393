#!/bin/bash
# per-kernel durations of the grouped relation call (rocprofv3 kernel trace): tools/rel_prof_grouped.sh "G MODE [LIB]" ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "$@"; do set -- $cfg; echo "== G=$1 HVR_REL_GROUPED=$2 lib=${3:-product}"; rm -rf /tmp/pr
  HVR_BENCH_LIB=${3:-} HVR_REL_GROUPED=$2 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pr -o rel -- python tools/rel_bench.py --groups $1 --iters 20 2>/dev/null | tail -1
  python tools/rocpd_stats.py $(find /tmp/pr -name "*.db" | head -1) | head -5 | tail -3 | cut -c1-60,100-170; done
