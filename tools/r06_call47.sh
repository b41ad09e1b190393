#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python bench.py --steps 20 --warmup 3 --tol-clips 48 --no-train-step --no-side-loops > gpurun_out/bench_tol48.json 2> gpurun_out/bench_tol48.err
python - <<'P'
import json
d=json.load(open('gpurun_out/bench_tol48.json'))
w=d.get('within_tolerance')
print('within_tolerance', None if w is None else {k: w[k] for k in ('dtype','frames_per_s','clips_checked','clips_passed','clips_with_the_oracles_proposal_lists','clips_proven_by_injection') if k in w})
if w: print(' box', w['per_clip']['max_box_err_vs_f32']); print(' ties', [(t['clip'], [(f['frame'], round(f['iou_f64'],7), f['is_tie']) for f in t['frames']]) for t in w['per_clip']['nms_threshold_ties']])
for r in d.get('within_tolerance_failed') or []:
    print(r['dtype'], 'passes', sum(r['passes']), 'of', len(r['passes']), 'failed clips', [i for i,p in enumerate(r['passes']) if not p])
    print('  box err', r['max_box_err_vs_f32'])
    print('  lists not equal', [i for i,e in enumerate(r['proposal_lists_equal_the_oracles']) if not e])
    print('  ties', [(t['clip'], [(f['frame'], f['iou_f64'], f['is_tie']) for f in t['frames']]) for t in r['nms_threshold_ties']])
    print('  injected', [(i, x['max_box_err'], x['within_tolerance']) for i,x in enumerate(r['with_the_oracles_proposals_injected']) if x])
P
