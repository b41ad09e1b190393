#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python tools/readout_probe.py --clip 1 --mode f16x2 --branch 0 2>&1 | grep -v amdgpu | cut -c1-330
timeout 600 python tools/readout_probe.py --clip 1 --mode f16x2 --branch 0 --one-level 2>&1 | grep -v amdgpu | cut -c1-330
