cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/profiles; mkdir -p $out
timeout 600 python bench.py --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<'P'
import json
d = json.loads(open('gpurun_out/profiles/bench.json').read().strip().split('\n')[-1])
print('value', d['value'], d['value_spread']['frames_per_s'], 'single', d['single_lane']['frames_per_s_per_gpu'])
print('roofline', {k: d['roofline'][k] for k in ('achieved', 'frac', 'traffic', 'traffic_source', 'avg_ms')})
print('stream', d['graphed_stream']['pipelined_window_cus']['window_on_confined_stream'], 'ladder best', d['precision_ladder']['fastest_mode_within_tolerance'])
P
