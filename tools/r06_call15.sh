#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c15; mkdir -p $O
for cfg in "4 5" "5 4" "4 4" "2 5" "10 2" "5 2" "4 3" "5 3"; do
  set -- $cfg
  timeout 300 python bench.py --steps 20 --warmup 5 --clips $1 --lanes $2 --no-cpu-baseline --no-train-step --no-side-loops --no-f32-leg > $O/b_$1_$2.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('$O/b_$1_$2.json'))
print('clips $1 lanes $2: value', d['value'], 'spread', d['value_spread']['frames_per_s'], 'batched eager', d.get('single_lane_batched',{}).get('frames_per_s'), 'roofline', d['roofline']['frac'])
PY
done
