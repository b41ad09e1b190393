#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c5; mkdir -p $O
timeout 600 python -m pytest tests/test_precision_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "big_tile or tile_shape or bottleneck or conv_on_half or gemm" > $O/pytest_bigtile.log 2>&1; tail -3 $O/pytest_bigtile.log
for pf in 1 0; do
  echo "== HVR_BIGTILE_PF=$pf" >> $O/bigtile_pf.txt
  HVR_BIGTILE_PF=$pf timeout 300 python tools/bigtile_probe.py --frames 15,60 --shapes reduce,res5reduce,fc1,qk >> $O/bigtile_pf.txt 2>&1
  HVR_BIGTILE_PF=$pf timeout 300 python tools/bigtile_probe.py --mode f16x2 --frames 30 --shapes reduce,res5reduce,fc1 >> $O/bigtile_pf.txt 2>&1
done
grep -v amdgpu $O/bigtile_pf.txt
