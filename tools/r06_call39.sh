#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/tol_clip_probe.py --clips 12 9 --mode f16x2 > gpurun_out/tol_probe_f16x2.txt 2>&1; tail -30 gpurun_out/tol_probe_f16x2.txt | cut -c1-330
timeout 900 python tools/tol_clip_probe.py --clips 13 17 20 --mode f32 > gpurun_out/tol_probe_f32.txt 2>&1; tail -30 gpurun_out/tol_probe_f32.txt | cut -c1-330
