cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/profiles; mkdir -p $out
timeout 700 python bench.py --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<'P'
import json
d = json.loads(open('gpurun_out/profiles/bench.json').read().strip().split('\n')[-1])
print('value', d['value'], d['value_spread']['frames_per_s'], 'single', d['single_lane']['frames_per_s_per_gpu'])
print('roofline', {k: d['roofline'][k] for k in ('achieved', 'frac', 'traffic', 'traffic_source', 'avg_ms')})
print('stream', d['graphed_stream']['pipelined_window_cus']['window_on_confined_stream'], 'ladder best', d['precision_ladder']['fastest_mode_within_tolerance'], 'f32 roofline', d['f32_parity_mode']['roofline']['frac'])
P
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_final.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_final.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke_final.log
