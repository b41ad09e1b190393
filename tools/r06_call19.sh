#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "hvr or hnmb or g15" > gpurun_out/t19.log 2>&1; tail -5 gpurun_out/t19.log
timeout 400 python tools/train_census.py --head hvr > gpurun_out/train_census_hvr.txt 2> gpurun_out/train_census_hvr.err
timeout 400 python tools/train_census.py --head selsa > gpurun_out/train_census_selsa.txt 2> gpurun_out/train_census_selsa.err
timeout 300 python tools/train_bench.py --steps 10 --warmup 3 --head hvr 2>&1 | tail -1
