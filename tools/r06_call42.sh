#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python bench.py --steps 20 --warmup 3 --tol-clips 24 --no-train-step --no-side-loops > gpurun_out/bench_tol24_b.json 2> gpurun_out/bench_tol24_b.err
python - <<'P'
import json
d=json.load(open('gpurun_out/bench_tol24_b.json'))
w=d.get('within_tolerance')
print('within_tolerance', None if w is None else {k: w[k] for k in ('dtype','frames_per_s','clips_checked','clips_passed','clips_with_the_oracles_proposal_lists','clips_proven_by_injection') if k in w})
rows = d.get('within_tolerance_failed') or []
if w: rows = rows + [dict(dtype=w['dtype'], **(w.get('per_clip') or {}))]
for r in rows:
    print(r['dtype'], 'passes', sum(r['passes']), 'of', len(r['passes']), [i for i,p in enumerate(r['passes']) if not p])
    print('  box err', r['max_box_err_vs_f32'])
    print('  lists equal', [i for i,e in enumerate(r['proposal_lists_equal_the_oracles']) if not e])
    print('  ties', [(t['clip'], [(f['frame'], round(f['iou_f64'],7) if f['iou_f64'] else None, f['is_tie']) for f in t['frames']]) for t in r['nms_threshold_ties']])
P
timeout 1500 python tools/noise_contrib.py --mode f16x2 --head selsa --clip 0 2>&1 | grep -E "everything|^rpn|backbone" 
