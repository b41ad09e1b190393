#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for h in hvr selsa; do
rm -rf /tmp/th_ks; timeout 420 rocprofv3 --kernel-trace --stats -d /tmp/th_ks -o train -- python tools/train_bench.py --steps 38 --warmup 2 --head $h > gpurun_out/train_prof_$h.json 2>/dev/null
timeout 200 python tools/rocpd_stats.py $(find /tmp/th_ks -name "*.db" | head -1) > gpurun_out/train_kernel_stats_${h}_40.txt
done
