#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c11; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "bottleneck or expand" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/l3_fused_bench.py > $O/l3_fused.txt 2>&1; grep -v amdgpu $O/l3_fused.txt
