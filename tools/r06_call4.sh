#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c4; mkdir -p $O
timeout 600 python -m pytest tests/test_precision_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "big_tile or tile_shape or conv_loader or bottleneck or conv_on_half" > $O/pytest_bigtile.log 2>&1; tail -5 $O/pytest_bigtile.log
echo "== persistent (256 workgroups)" > $O/bigtile_probe.txt
timeout 300 python tools/bigtile_probe.py --frames 15,30,60,120 --shapes reduce,c3,expand,res5c3,rpn >> $O/bigtile_probe.txt 2>&1
echo "== one tile per workgroup (HVR_BIGTILE_WGS=0)" >> $O/bigtile_probe.txt
HVR_BIGTILE_WGS=0 timeout 300 python tools/bigtile_probe.py --frames 15,30,60,120 --shapes reduce,c3,expand,res5c3,rpn >> $O/bigtile_probe.txt 2>&1
echo "== split half, persistent" >> $O/bigtile_probe.txt
timeout 300 python tools/bigtile_probe.py --mode f16x2 --frames 30,60 --shapes reduce,c3,expand,res5c3 >> $O/bigtile_probe.txt 2>&1
echo "== split half, one tile per workgroup" >> $O/bigtile_probe.txt
HVR_BIGTILE_WGS=0 timeout 300 python tools/bigtile_probe.py --mode f16x2 --frames 30,60 --shapes reduce,c3,expand,res5c3 >> $O/bigtile_probe.txt 2>&1
grep -v amdgpu $O/bigtile_probe.txt
HVR_BENCH_LIB=abtest/libhvr_bgclk.so timeout 300 python tools/bigtile_probe.py --clk --frames 60 --shapes reduce,c3 > $O/bigtile_clk.txt 2>&1
