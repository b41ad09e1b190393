cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/profiles; mkdir -p $out
timeout 420 bash tools/collect_traffic.sh > gpurun_out/collect_traffic_s9.log 2>&1; tail -3 gpurun_out/collect_traffic_s9.log
timeout 600 python bench.py --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<'P'
import json
d = json.loads(open('gpurun_out/profiles/bench.json').read().strip().split('\n')[-1])
print('value', d['value'], d['value_spread']['frames_per_s'], 'single', d['single_lane']['frames_per_s_per_gpu'])
print('roofline', {k: d['roofline'][k] for k in ('achieved', 'frac', 'traffic', 'avg_ms')})
print('stream', d['graphed_stream'].get('pipelined_window_cus'))
print('rpn', d['kernel_classes']['rpn_proposals'], 'key', d['kernel_classes']['relation_key'])
P
