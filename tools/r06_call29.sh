#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k parked > gpurun_out/t29.log 2>&1; grep -n "AssertionError\|assert " gpurun_out/t29.log | head; tail -5 gpurun_out/t29.log
