#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c8; mkdir -p $O
rm -rf /tmp/kp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kp -o k -- python tools/key_bench.py --groups 4 --iters 20 > /tmp/kp.log 2>&1
DB=$(find /tmp/kp -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > $O/key_kernel_stats.txt 2>&1 || (find /tmp/kp -name "*stats*" | head; tail -5 /tmp/kp.log)
head -20 $O/key_kernel_stats.txt | cut -c1-200
