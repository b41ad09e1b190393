"""Training-step throughput of SelsaRCNN on the HIP path (BASELINE.json configs[4]'s shape with the SELSA head: the reference's
tools/dist_train.sh iteration = forward_train + backward + gradient all-reduce + clip + SGD).

    python tools/train_bench.py [--steps K] [--warmup W] [--size H W] [--nms-post N]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py   # N replicas, RCCL all-reduce

One iteration = one training sample per rank = 1 key + 2 reference frames of 600x1000 (configs/faster_rcnn_r101_selsa_c5.py:
imgs_per_gpu = 1), 300 proposals per frame, loss-ranked second sampler keeping 128 rows; f32 master parameters, compute dtype bf16 (default: operands rounded to
bf16, f32 accumulation and weight gradients) or f32 (exact-f32 MFMA path, the parity mode).  Synthetic frames, random-init weights, four synthetic ground-truth boxes.  Prints one JSON line (rank 0)."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvrnet_amd  # noqa: E402
from hvrnet_amd import synthetic as S  # noqa: E402
from hvrnet_amd.config import hvr_train_config, selsa_train_config  # noqa: E402
from hvrnet_amd.dist_train import C4Prefetcher, FlatParams, train_detector_iteration  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--size', type=int, nargs=2, default=[600, 1000])
    ap.add_argument('--nms-post', type=int, default=300)
    ap.add_argument('--frames', type=int, default=3)
    ap.add_argument('--head', choices=['selsa', 'hvr'], default='selsa', help='selsa: SelsaRCNN, 1 key + 2 ref frames; hvr: HNMBRCNN, '
                    '5 videos x 3 frames in, 3 videos chosen (configs[4])')
    ap.add_argument('--no-prefetch', action='store_true', help='hvr: the frozen backbone in line with the step instead of one batch ahead on a second stream (dist_train.C4Prefetcher)')
    ap.add_argument('--overlap', action='store_true', help='conv weight gradients on a second HIP stream (train_ops.wgrad_overlap)')
    ap.add_argument('--cprofile', action='store_true', help='host-side cProfile of 5 iterations (stderr)')
    ap.add_argument('--detail', action='store_true', help='print the GEMM / conv calls of one iteration by shape (HIP-event times)')
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='bf16', help='compute dtype (parameters, gradients and the update stay f32)')
    args = ap.parse_args()
    rank, world, local = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    dev = 'cuda:%d' % local
    if world > 1:
        dist.init_process_group('nccl', device_id=torch.device(dev))
    hw = tuple(args.size)
    pad = tuple((v + 15) // 16 * 16 for v in hw)
    cdt = torch.float32 if args.dtype == 'f32' else torch.bfloat16
    sx, sy = hw[1] / 1000.0, hw[0] / 600.0
    gt_b = torch.tensor([[120., 80., 420., 330.], [296., 136., 359., 199.], [500., 100., 780., 300.], [820., 420., 865., 460.]])
    gt_b = (gt_b * torch.tensor([sx, sy, sx, sy])).to(dev)
    gt_l = torch.tensor([3, 17, 9, 22]).to(dev)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    if args.head == 'selsa':
        T = args.frames
        cfg = selsa_train_config(nms_post=args.nms_post, rcnn_sampler_num=128, t_dim=T)
        workload = '1 key + %d ref frames' % (T - 1)
    else:
        T = 15                                                  # 3 videos of the key class + 2 of other classes, 3 frames each
        cfg = hvr_train_config(nms_post=args.nms_post, rcnn_sampler_num=128)
        workload = '5 videos x 3 frames, 3 chosen'
    model = hvrnet_amd.enable_training(hvrnet_amd.build_model(cfg, S.synth_state_dict(args.head), cdt, dev))
    flat = FlatParams(model)
    imgs = torch.cat([S.synth_frame(1000 * rank + i, img_hw=hw, pad_hw=pad) for i in range(T)], 0).to(dev)
    metas = [S.synth_meta(hw, pad) for _ in range(T)]
    data = dict(img=imgs, img_meta=metas, return_loss=True, gt_bboxes=[gt_b] * T, gt_labels=[gt_l] * T, generator=gen)

    pre = C4Prefetcher(model) if (args.head == 'hvr' and not args.no_prefetch) else None
    if pre is not None:
        pre.start(imgs)

    def step():
        d = data
        if pre is not None:                  # this batch's C4 was computed while the previous batch trained; start the next batch's
            d = dict(data, c4=pre.take())
            pre.start(imgs)
        return train_detector_iteration(model, flat, d, lr=1e-4, momentum=0.9, weight_decay=1e-4, max_norm=35.0, overlap_wgrad=args.overlap)

    for _ in range(args.warmup):
        log = step()
    torch.cuda.synchronize()
    if args.detail and rank == 0:
        from hvrnet_amd import native
        native.profile_begin(('gemm', 'conv'), detail=True)
        step()
        prof = native.profile_end()
        tot = sum(d['ms'] for d in prof.values())
        print('# one iteration: %d gemm/conv calls, %.2f ms, %.1f TFLOP/s overall' % (sum(d['calls'] for d in prof.values()), tot,
                                                                                   sum(d['work'] for d in prof.values()) / tot / 1e9), file=sys.stderr)
        for tag, d in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])[:45]:
            print('# %-44s calls %3d  %8.3f ms  %7.1f TF/s' % (tag, d['calls'], d['ms'], d['work'] / d['ms'] / 1e9), file=sys.stderr)
    if args.cprofile and rank == 0:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats('tottime').print_stats(45)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        log = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)
    if rank == 0:
        print(json.dumps(dict(metric='%s training iterations/sec (%s, %dx%d, %d proposals)' % (args.head.upper(), workload, hw[1], hw[0], args.nms_post),
                              value=round(world * args.steps / dt, 3), unit='iterations/s', frames_per_s=round(world * args.steps * T / dt, 2),
                              n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * dt / args.steps, 2),
                              dtype=args.dtype, data='synthetic', params=int(flat.flat.numel()), backbone_prefetch=pre is not None,
                              last_losses={k: round(float(v), 4) for k, v in log.items()})))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
