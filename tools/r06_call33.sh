#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
if [ $v = 1 ]; then export HVR_DBG_X_NC4=1; else unset HVR_DBG_X_NC4; fi
echo "== NC4 forced: $v"
timeout 300 python tools/stream_bench.py --steps 60 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['window_on_confined_stream'], d['window_on_the_callers_stream'], d['graphs_alone'], d['same_detections'])"
done
unset HVR_DBG_X_NC4
timeout 300 python tools/frame_breakdown.py 2>/dev/null | head -8
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "expand or tail or bottleneck" 2>&1 | tail -2
for h in selsa; do timeout 300 python tools/train_bench.py --steps 20 --warmup 3 --head $h 2>&1 | tail -1 | cut -c150-260; done
