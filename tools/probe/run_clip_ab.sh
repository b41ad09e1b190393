cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in 4 15 4 15; do echo "== HVR_RPN_WIDE=$w"; HVR_RPN_WIDE=$w timeout 400 python bench.py --steps 20 --warmup 3 --repeats 3 --no-f32-leg --no-cpu-baseline --no-train-step --no-side-loops 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('value', d['value'], d['value_spread']['frames_per_s'], 'single_lane', d['single_lane']['frames_per_s_per_gpu'], d['single_lane']['regions'], 'rpn', d['kernel_classes']['rpn_proposals'], 'gstream', d.get('graphed_stream',{}).get('frames_per_s_per_gpu'))
"; done > gpurun_out/clip_ab.txt 2>&1
cat gpurun_out/clip_ab.txt
