#!/bin/bash
# re-entry check of the restored tree: GPU suite, then profile parts C (training, vendor calibration, ingest) and E (stream mode)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputest_r06_reentry.log 2>&1; tail -3 gpurun_out/gputest_r06_reentry.log
bash tools/collect_profiles.sh C
# HVR training step's kernel stats (no such breakdown in profiles/)
rm -rf /tmp/th_ks; timeout 420 rocprofv3 --kernel-trace --stats -d /tmp/th_ks -o train -- python tools/train_bench.py --steps 3 --warmup 1 --head hvr > /dev/null 2>&1
timeout 200 python tools/rocpd_stats.py $(find /tmp/th_ks -name "*.db" | head -1) > gpurun_out/profiles/train_kernel_stats_hvr.txt
bash tools/collect_profiles.sh E
