#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c12; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "bottleneck or expand" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
echo "== wave-pair form" > $O/l3_fused.txt
timeout 300 python tools/l3_fused_bench.py --frames 30,60 >> $O/l3_fused.txt 2>&1
echo "== NX = 16 instance (HVR_L3_PAIR=0)" >> $O/l3_fused.txt
HVR_L3_PAIR=0 timeout 300 python tools/l3_fused_bench.py --frames 30,60 >> $O/l3_fused.txt 2>&1
grep -v amdgpu $O/l3_fused.txt
