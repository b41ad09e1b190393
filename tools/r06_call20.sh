#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_targets_gpu.py tests/test_train_gpu.py -x -q > gpurun_out/t20.log 2>&1; tail -5 gpurun_out/t20.log
for h in hvr selsa; do
timeout 300 python tools/train_bench.py --steps 10 --warmup 2 --head $h 2>&1 | tail -1
done
timeout 400 python tools/train_census.py --head hvr > gpurun_out/train_census_hvr.txt 2> gpurun_out/train_census_hvr.err
rm -rf /tmp/th_ks; timeout 420 rocprofv3 --kernel-trace --stats -d /tmp/th_ks -o train -- python tools/train_bench.py --steps 3 --warmup 1 --head hvr > /dev/null 2>&1
timeout 200 python tools/rocpd_stats.py $(find /tmp/th_ks -name "*.db" | head -1) > gpurun_out/train_kernel_stats_hvr_b.txt
