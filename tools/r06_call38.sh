#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python bench.py --steps 20 --warmup 3 --tol-clips 24 --no-train-step --no-side-loops > gpurun_out/bench_tol24.json 2> gpurun_out/bench_tol24.err
python - <<'P'
import json
d=json.load(open('gpurun_out/bench_tol24.json'))
w=d.get('within_tolerance')
print('within_tolerance', None if w is None else {k: w[k] for k in ('dtype','frames_per_s','clips_checked','clips_passed','clips_with_the_oracles_proposal_lists','clips_proven_by_injection') if k in w})
if w: print(json.dumps(w.get('per_clip'))[:1500])
print('failed', d.get('within_tolerance_failed'))
P
