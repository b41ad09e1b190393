#!/bin/bash
# training-step throughput + per-kernel times; writes gpurun_out/train_bench.json and gpurun_out/train_kernel_stats.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/train_bench.py --steps 10 --warmup 2 "$@" > gpurun_out/train_bench.json 2> gpurun_out/train_bench.err
tail -2 gpurun_out/train_bench.err; cat gpurun_out/train_bench.json
rm -rf /tmp/tp; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/tp -o r -- python tools/train_bench.py --steps 3 --warmup 1 "$@" > /tmp/tp.log 2>&1
DB=$(find /tmp/tp -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > gpurun_out/train_kernel_stats.txt && head -40 gpurun_out/train_kernel_stats.txt | cut -c1-200 || tail -5 /tmp/tp.log
