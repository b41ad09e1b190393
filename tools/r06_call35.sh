#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "batched or unpack_conv_wgrads" 2>&1 | tail -5
