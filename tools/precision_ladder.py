"""The precision ladder at full size: one 15-frame window of configs[2] (HVR; --head selsa for configs[1]) through the HIP path
in every compute mode -- bf16, half, split half, exact f32, and per-module mixes -- against oracle.clip_forward on the same
synthetic frames: where (C4 map, proposal lists, head, read-out) a mode leaves north_star's tolerance (class indices exact,
scores / boxes within 1e-3), and what a window costs in that mode.  Prints one JSON object; bench.py reports the same table
(`precision_ladder`) for the modes it times.

  python tools/precision_ladder.py [--head hvr] [--modes bf16,f16,f16x2,f32,trunk_f16x2+head_f16] [--iters 5] [--out file.json]

The reference computes in f32 everywhere (configs/faster_rcnn_r101_hrnmp_c5.py has no fp16 key); its optional half islands
are mmdet/core/fp16/decorators.py:9-160, the read-out being force_fp32 (bbox_head.py:98,132) as it is here in every mode.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import hvrnet_amd  # noqa: E402
from hvrnet_amd import native, parity, synthetic as S  # noqa: E402
from hvrnet_amd.backbone import set_compute_dtype  # noqa: E402
from hvrnet_amd.config import hvr_config, selsa_config  # noqa: E402

DT = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f16x2': native.SPLIT, 'f32': torch.float32}


def apply_mode(model, mode):
    """'bf16' | 'f16' | 'f16x2' | 'f32', or a mix 'trunk_X+head_Y' (trunk = backbone + res5 + RPN head, head = RoI head)."""
    if mode in DT:
        set_compute_dtype(model, DT[mode])
        return
    parts = dict(p.split('_', 1) for p in mode.split('+'))
    set_compute_dtype(model, DT[parts.get('head', 'bf16')])
    for name in ('backbone', 'shared_head', 'rpn_head'):
        sub = getattr(model, name, None)
        if sub is not None:
            set_compute_dtype(sub, DT[parts['trunk']])
    if 'rpn' in parts:
        set_compute_dtype(model.rpn_head, DT[parts['rpn']])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--head', default='hvr')
    ap.add_argument('--modes', default='bf16,f16,f16x2,f32')
    ap.add_argument('--frames', type=int, default=15)
    ap.add_argument('--proposals', type=int, default=300)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    T, N, KEY = args.frames, args.proposals, args.frames // 2
    dev = 'cuda:0'
    import subprocess
    if not os.path.exists(os.path.join(ROOT, 'oracle', 'libhvr_oracle.so')):
        subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle')], check=True)
    from oracle import hvr_oracle as O

    frames = [S.synth_frame(i) for i in range(T)]
    metas = [S.synth_meta() for _ in range(T)]
    sd = S.synth_state_dict(args.head)
    t0 = time.time()
    with torch.no_grad():
        c4_ref = [O.resnet_c4(f, sd) for f in frames]
        res, inter = O.window_forward(c4_ref, metas, sd, args.head, KEY, N, T, rpn_cfg=dict(O.RPN_TEST_CFG, nms_post=N, max_num=N),
                                      return_intermediates=True)
    want = res if args.head == 'hvr' else [res[0]]
    oracle_s = time.time() - t0
    c4_ref = torch.cat(c4_ref, 0)
    c4_scale = c4_ref.abs().max().item()
    make = hvr_config if args.head == 'hvr' else selsa_config
    fr = torch.cat(frames, 0).to(dev)
    rows = []
    for mode in args.modes.split(','):
        model = hvrnet_amd.build_model(make(frame_interval=KEY, nms_post=N), sd, None, dev)
        apply_mode(model, mode)
        with torch.no_grad():
            c4 = model(img=fr, img_meta=metas, backbone_feat=True)[0]
            c4f = native.cast(c4.permute(0, 2, 3, 1), torch.float32).permute(0, 3, 1, 2).cpu() if c4.dtype != torch.float32 else c4.cpu()
            w = model.window_tensors(c4, metas)
            got = model(x=c4, img=None, img_meta=metas, forward_feat=True, return_loss=False, rescale=True)
            props = [p.to(dev) for p in inter['proposals']]
            got_inj = model(x=c4, img=None, img_meta=metas, proposals=props, forward_feat=True, return_loss=False, rescale=True)
        got = got if args.head == 'hvr' else [got]
        got_inj = got_inj if args.head == 'hvr' else [got_inj]
        same_lists, list_diffs = 0, []
        for i in range(T):
            a, b = w['proposals'][i].cpu(), inter['proposals'][i]
            if a.shape == b.shape and (a[:, :4] - b[:, :4]).abs().max().item() <= 2e-2 and (a[:, 4] - b[:, 4]).abs().max().item() <= 1e-3:
                same_lists += 1
            elif a.shape == b.shape:
                bad = ((a - b).abs().max(dim=1).values > 2e-2).nonzero().reshape(-1).tolist()
                list_diffs.append(dict(frame=i, rows=bad[:6], n_rows=len(bad), max_box=float((a[:, :4] - b[:, :4]).abs().max()),
                                       max_score=float((a[:, 4] - b[:, 4]).abs().max()),
                                       first=[[round(v, 4) for v in a[bad[0]].tolist()], [round(v, 4) for v in b[bad[0]].tolist()]] if bad else None))
        po = parity.proposal_overlap([p.cpu().numpy() for p in w['proposals']], [p.numpy() for p in inter['proposals']])
        st = [parity.strict(g, r) for g, r in zip(got, want)]
        st_inj = [parity.strict(g, r) for g, r in zip(got_inj, want)]
        tr = parity.track(got[-1], want[-1])
        # cost of one window in this mode (eager, one lane; everything recomputed)
        torch.cuda.synchronize()
        ts = []
        with torch.no_grad():
            for _ in range(args.iters + 1):
                t1 = time.time()
                c4b = model(img=fr, img_meta=metas, backbone_feat=True)[0]
                model(x=c4b, img=None, img_meta=metas, forward_feat=True, return_loss=False, rescale=True)
                torch.cuda.synchronize()
                ts.append(time.time() - t1)
        ms = 1e3 * sorted(ts[1:])[len(ts[1:]) // 2]
        row = dict(mode=mode, window_ms=round(ms, 3), frames_per_s=round(1e3 / ms, 2),
                   c4_rel_err=float((c4f - c4_ref).abs().max().item() / c4_scale),
                   proposal_lists_equal='%d/%d' % (same_lists, T), proposal_list_diffs=list_diffs[:3], proposal_overlap_mean=round(po['mean'], 4),
                   class_flips=[s['class_flips'] for s in st], detections=[s['n'] for s in st],
                   max_score_err=max(s['max_score_err'] for s in st), max_box_err_px=max(s['max_box_err'] for s in st),
                   tracked=dict(matched='%d/%d' % (tr['matched'], tr['n_ref']), max_score_err=tr['max_score_err'], max_box_err_px=tr['max_box_err']),
                   with_oracle_proposals=dict(class_flips=[s['class_flips'] for s in st_inj], max_score_err=max(s['max_score_err'] for s in st_inj),
                                              max_box_err_px=max(s['max_box_err'] for s in st_inj)))
        # north_star's bar: the ONE definition in hvrnet_amd/parity.py (classes exact, scores < 1e-3, boxes < 1e-3 px + 1.3e-6 x extent)
        row['within_tolerance'] = all(parity.within_tolerance(s_) for s_ in st)
        row['fixed_bar_r04'] = all(parity.fixed_bar_r04(s_) for s_ in st)
        rows.append(row)
        print(json.dumps(row), flush=True)
        del model
        torch.cuda.empty_cache()
    out = dict(head=args.head, T=T, N=N, oracle_seconds=round(oracle_s, 1), c4_scale=c4_scale, device=torch.cuda.get_device_name(0), rows=rows)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, 'w') as f:
            json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
