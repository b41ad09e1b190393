#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for h in selsa hvr; do
rm -rf /tmp/th_ks; timeout 420 rocprofv3 --kernel-trace -d /tmp/th_ks -o train -- python tools/train_bench.py --steps 20 --warmup 3 --head $h > gpurun_out/train_prof_$h.json 2>/dev/null
db=$(find /tmp/th_ks -name "*.db" | head -1)
echo "== $h"; tail -1 gpurun_out/train_prof_$h.json | cut -c1-260
timeout 200 python tools/gpu_idle.py $db stem_fused_kernel 8 8 > gpurun_out/train_gpu_idle_$h.txt 2>&1; head -9 gpurun_out/train_gpu_idle_$h.txt
done
