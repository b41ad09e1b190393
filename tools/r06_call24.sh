#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/train_bench.py --steps 5 --warmup 3 --head hvr --cprofile > gpurun_out/train_cprofile_hvr.txt 2>&1
timeout 300 python tools/train_bench.py --steps 5 --warmup 3 --head selsa --cprofile > gpurun_out/train_cprofile_selsa.txt 2>&1
python - <<'P'
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import hvrnet_amd
from hvrnet_amd import synthetic as S
from hvrnet_amd.config import hvr_train_config, selsa_train_config
from hvrnet_amd.dist_train import FlatParams, train_detector_iteration
dev='cuda:0'
for head in ('hvr','selsa'):
    T = 15 if head=='hvr' else 3
    cfg = hvr_train_config(nms_post=300, rcnn_sampler_num=128) if head=='hvr' else selsa_train_config(nms_post=300, rcnn_sampler_num=128, t_dim=3)
    model = hvrnet_amd.enable_training(hvrnet_amd.build_model(cfg, S.synth_state_dict(head), torch.bfloat16, dev))
    flat = FlatParams(model)
    hw, pad = (600,1000),(608,1008)
    imgs = torch.cat([S.synth_frame(i, img_hw=hw, pad_hw=pad) for i in range(T)], 0).to(dev)
    metas = [S.synth_meta(hw, pad) for _ in range(T)]
    gt_b = torch.tensor([[120., 80., 420., 330.], [296., 136., 359., 199.], [500., 100., 780., 300.], [820., 420., 865., 460.]]).to(dev)
    gt_l = torch.tensor([3, 17, 9, 22]).to(dev)
    gen = torch.Generator(device=dev).manual_seed(1234)
    data = dict(img=imgs, img_meta=metas, return_loss=True, gt_bboxes=[gt_b]*T, gt_labels=[gt_l]*T, generator=gen)
    for _ in range(4): train_detector_iteration(model, flat, data, lr=1e-4)
    torch.cuda.synchronize()
    hs, ts = [], []
    for _ in range(8):
        t0=time.perf_counter(); train_detector_iteration(model, flat, data, lr=1e-4); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
        hs.append((t1-t0)*1e3); ts.append((t2-t0)*1e3)
    print(head, 'host return ms', ['%.2f'%v for v in hs], 'with sync', ['%.2f'%v for v in ts])
P
