#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c3; mkdir -p $O
timeout 300 python tools/noise_contrib.py --mode f16x2 --head selsa --clip 0 > $O/noise_contrib_f16x2_selsa.txt 2>&1
tail -9 $O/noise_contrib_f16x2_selsa.txt
timeout 300 python tools/noise_contrib.py --mode f16x2 --head hvr --clip 7 > $O/noise_contrib_f16x2_hvr_clip7.txt 2>&1
tail -9 $O/noise_contrib_f16x2_hvr_clip7.txt
