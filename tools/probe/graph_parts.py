import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tools'))
import hvrnet_amd
from hvrnet_amd import native, synthetic as S
from hvrnet_amd.config import hvr_config
from precision_ladder import apply_mode
T, N, dev = 15, 300, 'cuda:0'
mode = sys.argv[1]
model = hvrnet_amd.build_model(hvr_config(frame_interval=T // 2, nms_post=N), S.synth_state_dict('hvr'), None, dev)
apply_mode(model, mode)
fr = torch.cat([S.synth_frame(i) for i in range(T)], 0).to(dev)
metas = [S.synth_meta() for _ in range(T)]
with torch.no_grad():
    c4_ref = model(img=fr, img_meta=metas, backbone_feat=True)[0].clone()
    w_ref = model.window_tensors(c4_ref, metas, speculate=True)
    roi_ref = w_ref['roi_feats'].clone()
    f1_ref = model.bbox_head.fc1_rows(roi_ref).clone()
torch.cuda.synchronize()


def capture(fn):
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.no_grad(), torch.cuda.stream(st):
        for _ in range(2):
            out = fn()
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            out = fn()
    torch.cuda.current_stream().wait_stream(st)
    return g, out


def run2(name, fn, ref, pick=lambda o: o):
    lanes = [torch.cuda.Stream() for _ in range(2)]
    gs = [capture(fn) for _ in range(2)]
    torch.cuda.synchronize()
    bad = 0
    for it in range(6):
        for k in range(2):
            with torch.cuda.stream(lanes[k]):
                gs[k][0].replay()
        torch.cuda.synchronize()
        for k in range(2):
            bad += 0 if torch.equal(pick(gs[k][1]), ref) else 1
    print('%-40s %s: %d of 12 concurrent replays differ' % (name, mode, bad), flush=True)


run2('backbone (img -> C4)', lambda: model(img=fr, img_meta=metas, backbone_feat=True)[0], c4_ref)
run2('res5 + RPN + RoIAlign (C4 -> roi feats)', lambda: model.window_tensors(c4_ref, metas, speculate=True)['roi_feats'], roi_ref)
run2('fc_new_1 rows', lambda: model.bbox_head.fc1_rows(roi_ref), f1_ref)
cur = dict(start=(T // 2) * N, length=N)
with torch.no_grad():
    h_ref = model.bbox_head.forward_from_f1(f1_ref, [cur])
run2('relation head (f1 -> logits)', lambda: model.bbox_head.forward_from_f1(f1_ref, [cur])[0][1], h_ref[0][1].clone())
p = model.bbox_head.packed(f1_ref.device)
D = 1024
with torch.no_grad():
    qk = native.gemm(f1_ref, p['wqk1'], p['bqk1'])
    q, kk = qk[:, :D], qk[:, D:]
    o_ref = native.relation_fwd(q, kk, f1_ref, 1 / 32.).clone()
    ok_ref = native.relation_fwd(q[2100:2400], kk, f1_ref, 1 / 32.).clone()
    g_ref = native.gemm(o_ref, p['wz1'], p['bz1'], resid=f1_ref, relu=True).clone()
    qk_ref = qk.clone()
torch.cuda.synchronize()
run2('gemm qk', lambda: native.gemm(f1_ref, p['wqk1'], p['bqk1']), qk_ref)
run2('relation full', lambda: native.relation_fwd(q, kk, f1_ref, 1 / 32.), o_ref)
run2('relation key rows', lambda: native.relation_fwd(q[2100:2400], kk, f1_ref, 1 / 32.), ok_ref)
run2('gemm out + resid', lambda: native.gemm(o_ref, p['wz1'], p['bz1'], resid=f1_ref, relu=True), g_ref)
print({k: (v.data_ptr(), v.numel()) for k, v in native._ws_cache.items() if k[0] == 'relation'})
with torch.no_grad():
    P_ref = native.relation_probs(q, kk, 1 / 32. / 256.).clone()
    vt_ref = native.transpose_pad(f1_ref, native.relation_ldp(4500)).clone()
    o2_ref = native.gemm(P_ref, vt_ref, alpha=1.0).clone()
torch.cuda.synchronize()
run2('probs (scores + normalise)', lambda: native.relation_probs(q, kk, 1 / 32. / 256.), P_ref)
run2('V transpose', lambda: native.transpose_pad(f1_ref, native.relation_ldp(4500)), vt_ref)
run2('apply GEMM P . Vt', lambda: native.gemm(P_ref, vt_ref, alpha=1.0), o2_ref)
for hint in (1, 4, 5, 11, 12):
    with torch.no_grad():
        r_ = native.gemm(P_ref, vt_ref, alpha=1.0, tile=hint).clone()
    torch.cuda.synchronize()
    run2('apply GEMM hint %d' % hint, lambda: native.gemm(P_ref, vt_ref, alpha=1.0, tile=hint), r_)
