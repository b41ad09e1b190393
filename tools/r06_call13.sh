#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c13; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "expand or bottleneck" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
echo "== 256-pixel panels (default)" > $O/expand_wide.txt
timeout 300 python tools/expand_bench.py --stages l3 --modes bf16 --frames 60,120 >> $O/expand_wide.txt 2>&1
echo "== 128-pixel panels (HVR_EXPAND_WIDE=0)" >> $O/expand_wide.txt
HVR_EXPAND_WIDE=0 timeout 300 python tools/expand_bench.py --stages l3 --modes bf16 --frames 60,120 >> $O/expand_wide.txt 2>&1
grep -v amdgpu $O/expand_wide.txt
