#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c7; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_precision_gpu.py -x -q -m gpu -k "relation or gemm or tile_shape" > $O/pytest_rel.log 2>&1; tail -3 $O/pytest_rel.log
timeout 200 python tools/key_bench.py --groups 4 > $O/key_bench.txt 2>&1; grep -v amdgpu $O/key_bench.txt
timeout 600 python -m pytest tests -x -q -m gpu -k "batched_clips or graphs or grouped or head" > $O/pytest_heads.log 2>&1; tail -3 $O/pytest_heads.log
