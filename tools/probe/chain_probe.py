"""Which of the two chains behind C4 ends last: the RPN (second stream) or res5 (main stream)?"""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import hvrnet_amd
from hvrnet_amd import synthetic as S
from hvrnet_amd.config import hvr_config
dev = torch.device('cuda', 0)
T = 15
model = hvrnet_amd.build_model(hvr_config(frame_interval=T // 2, nms_post=300), S.synth_state_dict('hvr'), torch.bfloat16, dev)
frames = torch.cat([S.synth_frame(i) for i in range(T)], 0).to(dev)
metas = [S.synth_meta() for _ in range(T)]
with torch.no_grad():
    c4 = model(img=frames, img_meta=metas, backbone_feat=True)[0]
    xc = model._cat_frames(c4)
    for it in range(6):
        torch.cuda.synchronize()
        main, side = torch.cuda.current_stream(dev), model._side_stream(dev)
        ev = lambda: torch.cuda.Event(enable_timing=True)
        ready, rpn_conv, done, res5 = ev(), ev(), ev(), ev()
        ready.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            rpn_outs = model.rpn_head([xc])
            rpn_conv.record(side)
            props, counts = model.rpn_head.get_bboxes_batched(rpn_outs[0], rpn_outs[1], metas, model.test_cfg.rpn)
            done.record(side)
        feats = model.shared_head(xc)
        res5.record(main)
        torch.cuda.synchronize()
        if it >= 2:
            print('RPN convs %.3f ms, RPN chain %.3f ms; res5 chain %.3f ms' % (ready.elapsed_time(rpn_conv), ready.elapsed_time(done), ready.elapsed_time(res5)))
    # each chain alone
    for name, fn in (('rpn alone', lambda: model.rpn_head.get_bboxes_batched(*model.rpn_head([xc])[:2], metas, model.test_cfg.rpn)), ('res5 alone', lambda: model.shared_head(xc))):
        for _ in range(2): fn()
        torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        print('%s %.3f ms' % (name, a.elapsed_time(b)))
