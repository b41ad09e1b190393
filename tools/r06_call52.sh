#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | grep -E "^E  |FAILED|assert|Error" | head -12
