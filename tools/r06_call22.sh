#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for h in hvr selsa; do
timeout 400 python tools/train_census.py --head $h --top 100 > gpurun_out/train_census_$h.txt 2> gpurun_out/train_census_$h.err
done
