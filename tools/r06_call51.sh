#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -x -q 2>&1 | tail -3
for h in hvr selsa hvr selsa; do timeout 300 python tools/train_bench.py --steps 20 --warmup 3 --head $h 2>&1 | tail -1 | cut -c1-30,150-240; done
rm -rf /tmp/th_ks; timeout 420 rocprofv3 --kernel-trace --stats -d /tmp/th_ks -o train -- python tools/train_bench.py --steps 10 --warmup 2 --head selsa > /dev/null 2>&1
timeout 200 python tools/rocpd_stats.py $(find /tmp/th_ks -name "*.db" | head -1) | grep -i "pack_conv\|unpack"
