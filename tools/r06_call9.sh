#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c9; mkdir -p $O
timeout 600 python -m pytest tests/test_precision_gpu.py -x -q -m gpu -k "relation or key_stage" > $O/pytest_rel.log 2>&1; tail -3 $O/pytest_rel.log
timeout 200 python tools/key_bench.py --groups 4 --dtype f16x2 > $O/key_bench_f16x2.txt 2>&1; grep -v amdgpu $O/key_bench_f16x2.txt
timeout 200 python tools/key_bench.py --groups 4 > $O/key_bench_bf16.txt 2>&1; grep -v amdgpu $O/key_bench_bf16.txt
