#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "prefetched" 2>&1 | tail -5
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_parity_gpu.py::test_prefetched_frozen_backbone_equals_the_in_line_pass > gpurun_out/gputest_r06_b.log 2>&1; tail -3 gpurun_out/gputest_r06_b.log
