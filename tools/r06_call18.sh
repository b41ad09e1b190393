#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python tools/train_census.py --head hvr > gpurun_out/train_census_hvr.txt 2> gpurun_out/train_census_hvr.err
timeout 400 python tools/train_census.py --head selsa > gpurun_out/train_census_selsa.txt 2> gpurun_out/train_census_selsa.err
timeout 300 python tools/train_bench.py --steps 5 --warmup 2 --head hvr --detail > gpurun_out/train_detail_hvr.txt 2>&1
timeout 300 python tools/train_bench.py --steps 5 --warmup 2 --head selsa --detail > gpurun_out/train_detail_selsa.txt 2>&1
tail -3 gpurun_out/train_census_hvr.err
