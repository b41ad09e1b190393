#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python tools/noise_contrib.py --mode f16x2 --head hvr --clip 7 --rpn-two-level > gpurun_out/noise_two_level_hvr7.txt 2>&1; grep -v amdgpu gpurun_out/noise_two_level_hvr7.txt | cut -c1-220
timeout 1500 python tools/noise_contrib.py --mode f16x2 --head selsa --clip 0 --rpn-two-level > gpurun_out/noise_two_level_selsa0.txt 2>&1; grep -v amdgpu gpurun_out/noise_two_level_selsa0.txt | cut -c1-220
