"""Training-target generation on the HIP path (hvrnet_amd/targets.py over hvr_max_iou_assign / hvr_sample_pos_neg /
hvr_box_targets / hvr_rpn_loss / hvr_ce_rows / hvr_det_loss_sampled) against G12 = the reference's own MaxIoUAssigner,
RandomSampler, OHEMHNLSampler, anchor_target, bbox_target, RPNHead.loss and BBoxHead.loss, and against the oracle on
seeded cases.  Integer results (assignments, sampled index sets, labels, weights) and IoUs are compared exactly; the
deltas go through logf (1e-6), the losses through a different summation order (1e-5).

The reference draws its random subsets with a host-side numpy shuffle; the HIP sampler takes one key per box and keeps
the smallest.  `_replay_keys` turns the subset the reference drew (stored in the fixture) into keys, so every step after
the draw is compared on identical samples."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import hvrnet_amd  # noqa: E402
from hvrnet_amd import native, targets as T, train_ops as TO  # noqa: E402
from hvrnet_amd.box_ops import AnchorGenerator  # noqa: E402
from tests.golden import cases as C  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def O():
    import subprocess
    if not os.path.exists(os.path.join(ROOT, 'oracle', 'libhvr_oracle.so')):
        subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle')], check=True)
    from oracle import hvr_oracle
    return hvr_oracle


@pytest.fixture(scope='module')
def g12():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'g12_targets.npz'))


def _replay_keys(n, chosen):
    keys = torch.ones(n)
    keys[torch.as_tensor(np.asarray(chosen)).long()] = 0.0
    return keys.to(DEV)


def _eq(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return np.array_equal(a, np.asarray(b))


def close(a, b, rtol, atol):
    torch.testing.assert_close(a.detach().float().cpu(), torch.as_tensor(np.asarray(b)).float(), rtol=rtol, atol=atol)


def _anchors():
    gen = AnchorGenerator(16, [4, 8, 16, 32], [0.5, 1.0, 2.0])
    return gen.grid_anchors((38, 63), 16, device=DEV).contiguous()


def _cfg(d):
    d = dict(d)
    d['assigner'] = dict(type='MaxIoUAssigner', ignore_iof_thr=-1, **d['assigner'])
    d['sampler'] = dict(type='RandomSampler', **d['sampler'])
    return d


META = dict(img_shape=(600, 1000, 3), pad_shape=(608, 1008, 3), scale_factor=1.0, flip=False)


def test_max_iou_assigner_matches_reference(g12):
    tc = C.target_case()
    anchors = _anchors()
    inside = torch.as_tensor(g12['inside']).to(DEV)
    assert _eq(T.anchor_inside_flags(anchors, None, (600, 1000, 3), 0).bool(), g12['inside'])
    asg = T.build_assigner(_cfg(C.RPN_TRAIN_CFG)['assigner'])
    # (a) compacted anchors, as the reference calls it
    res = asg.assign(anchors[inside].contiguous(), tc['gt_bboxes'].to(DEV))
    assert _eq(res.gt_inds, g12['rpn_gt_inds']) and _eq(res.max_overlaps, g12['rpn_max_overlaps'])
    # (b) all anchors + validity mask: same assignment at the inside anchors, -1 / -1 elsewhere
    res = asg.assign(anchors, tc['gt_bboxes'].to(DEV), valid=inside)
    assert _eq(res.gt_inds[inside], g12['rpn_gt_inds']) and _eq(res.max_overlaps[inside], g12['rpn_max_overlaps'])
    assert bool((res.gt_inds[~inside] == -1).all()) and bool((res.max_overlaps[~inside] == -1).all())
    # (c) proposals [n,5] with labels (the RCNN assigner)
    asg2 = T.build_assigner(_cfg(C.RCNN_TRAIN_CFG)['assigner'])
    res2 = asg2.assign(tc['proposals'].to(DEV), tc['gt_bboxes'].to(DEV), None, tc['gt_labels'].to(DEV))
    assert _eq(res2.gt_inds, g12['rcnn_gt_inds']) and _eq(res2.max_overlaps, g12['rcnn_max_overlaps'])
    want_labels = np.where(g12['rcnn_gt_inds'] > 0, tc['gt_labels'].numpy()[np.maximum(g12['rcnn_gt_inds'] - 1, 0)], 0)
    assert _eq(res2.labels, want_labels)
    with pytest.raises(ValueError):
        asg.assign(anchors, tc['gt_bboxes'][:0].to(DEV))


@pytest.mark.parametrize('n,k,seed', [(1, 1, 0), (63, 3, 1), (64, 1, 2), (300, 7, 3), (4097, 40, 4), (28728, 256, 5)])
def test_max_iou_assigner_matches_oracle_on_random_boxes(O, n, k, seed):
    """Boxes on an integer grid so equal IoUs (the `== gt maximum` rule, several boxes per gt) and exact thresholds occur."""
    g = torch.Generator().manual_seed(seed)
    xy = torch.randint(0, 40, (n, 2), generator=g).float() * 8
    wh = torch.randint(1, 6, (n, 2), generator=g).float() * 16 - 1
    boxes = torch.cat([xy, xy + wh], 1)
    gxy = torch.randint(0, 40, (k, 2), generator=g).float() * 8
    gwh = torch.randint(1, 6, (k, 2), generator=g).float() * 16 - 1
    gts = torch.cat([gxy, gxy + gwh], 1)
    for pos, neg, mn in ((0.7, 0.3, 0.3), (0.5, 0.5, 0.5), (0.5, (0.1, 0.5), 0.0)):
        if isinstance(neg, tuple):
            want_inds, want_ov = O.max_iou_assign(boxes, gts, pos, 2.0, mn)   # reproduce the tuple rule by hand below
            ov = O.bbox_overlaps(gts, boxes)
            mx, arg = ov.max(0)
            want_inds = torch.full((n,), -1, dtype=torch.long)
            want_inds[(mx >= neg[0]) & (mx < neg[1])] = 0
            want_inds[mx >= pos] = arg[mx >= pos] + 1
            gmx = ov.max(1)[0]
            for i in range(k):
                if gmx[i] >= mn:
                    want_inds[ov[i] == gmx[i]] = i + 1
        else:
            want_inds, want_ov = O.max_iou_assign(boxes, gts, pos, neg, mn)
        got_inds, got_ov = native.max_iou_assign(boxes.to(DEV), gts.to(DEV), pos, neg, mn)
        assert _eq(got_ov, want_ov.numpy()), (n, k, pos)
        assert _eq(got_inds, want_inds.numpy()), (n, k, pos)


@pytest.mark.parametrize('n,num,frac,ub,seed', [(10, 6, 0.5, -1, 0), (1000, 256, 0.5, -1, 1), (1000, 256, 0.5, 2.0, 2),
                                                (28728, 256, 0.5, -1, 3), (305, 300, 0.25, -1, 4), (2048, 128, 0.25, -1, 5),
                                                (5000, 512, 0.5, 0.0, 6)])
def test_sampler_matches_oracle(O, n, num, frac, ub, seed):
    """Continuous keys (no ties), coarse keys (many ties: lower index wins) and negative keys (the OHEM use: -loss)."""
    g = torch.Generator().manual_seed(seed)
    for kind in ('uniform', 'coarse', 'negative'):
        cls = torch.randint(-1, 3, (n,), generator=g)
        if seed % 2:
            cls[torch.rand(n, generator=g) < 0.9] = 0   # few positives: "take them all" branch
        keys = torch.rand(n, generator=g)
        if kind == 'coarse':
            keys = (keys * 4).floor() / 4
        if kind == 'negative':
            keys = -(keys * 8).floor() / 3 + 1
        pos, neg = O.sample_pos_neg(cls, keys, num, frac, ub)
        inds, counts = native.sample_pos_neg(cls.to(DEV), keys.to(DEV), num, int(num * frac), ub)
        c = counts.tolist()
        assert c == [pos.numel(), neg.numel()], (kind, c)
        assert inds[:c[0]].tolist() == pos.tolist(), kind
        assert inds[c[0]:c[0] + c[1]].tolist() == neg.tolist(), kind


def test_random_sampler_with_drawn_keys_is_a_valid_uniform_sample():
    """Without keys the sampler draws them: sizes as BaseSampler.sample prescribes, members from the right class, sorted,
    different draws differ, and over many draws every candidate is picked about equally often."""
    n = 2000
    gt_inds = torch.zeros(n, dtype=torch.long, device=DEV)
    gt_inds[::10] = 1            # 200 positives
    gt_inds[5::10] = -1          # 200 ignored
    boxes = torch.rand(n, 4, device=DEV)
    sampler = T.RandomSampler(num=256, pos_fraction=0.25, add_gt_as_proposals=False)
    gen = torch.Generator(device=DEV).manual_seed(7)
    hits = torch.zeros(n, device=DEV)
    first = None
    for it in range(200):
        res = sampler.sample(T.AssignResult(1, gt_inds, torch.zeros(n, device=DEV)), boxes, boxes[:1], generator=gen)
        pos, neg = res.pos_inds, res.neg_inds
        assert pos.numel() == 64 and neg.numel() == 192
        assert bool((gt_inds[pos] > 0).all()) and bool((gt_inds[neg] == 0).all())
        assert bool((pos[1:] > pos[:-1]).all()) and bool((neg[1:] > neg[:-1]).all())
        hits[pos] += 1
        hits[neg] += 1
        if first is None:
            first = pos.clone()
        elif it == 1:
            assert not torch.equal(first, pos)
    p, q = hits[gt_inds > 0] / 200, hits[gt_inds == 0] / 200
    assert abs(float(p.mean()) - 64 / 200) < 1e-6 and float((p - 0.32).abs().max()) < 0.15
    assert abs(float(q.mean()) - 192 / 1600) < 1e-6 and float((q - 0.12).abs().max()) < 0.10


def test_anchor_target_and_rpn_loss_match_reference(g12):
    tc = C.target_case()
    anchors = _anchors()
    keys = _replay_keys(anchors.shape[0], np.nonzero(g12['rpn_label_weights'] > 0)[0])
    cfg = _cfg(C.RPN_TRAIN_CFG)
    labels, lw, bt, bw, npos, nneg = T.anchor_target([[anchors]], None, [tc['gt_bboxes'].to(DEV)], [META], [0., 0., 0., 0.],
                                                     [1., 1., 1., 1.], cfg, keys_list=[keys])
    assert (npos, nneg) == (int(g12['rpn_num_pos']), int(g12['rpn_num_neg']))
    assert _eq(labels[0][0], g12['rpn_labels']) and _eq(lw[0][0], g12['rpn_label_weights']) and _eq(bw[0][0], g12['rpn_bbox_weights'])
    close(bt[0][0], g12['rpn_bbox_targets'], 1e-6, 1e-6)
    # loss + gradient on the fused head layout [H*W, 5A (+ pad)]
    A = 12
    o = torch.zeros((38 * 63, 64))
    o[:, :A] = tc['rpn_cls'][0].permute(1, 2, 0).reshape(-1, A)
    o[:, A:5 * A] = tc['rpn_reg'][0].permute(1, 2, 0).reshape(-1, 4 * A)
    o = o.to(DEV).requires_grad_(True)
    single = T.anchor_target_single(anchors, None, tc['gt_bboxes'].to(DEV), META, [0.] * 4, [1.] * 4, cfg, keys=keys)
    losses = TO.rpn_loss(o, A, *single[:4], single[4], beta=1.0 / 9.0)
    close(losses['loss_rpn_cls'], g12['loss_rpn_cls'], 1e-5, 1e-6)
    close(losses['loss_rpn_bbox'], g12['loss_rpn_bbox'], 1e-5, 1e-6)
    losses['total'].sum().backward()
    d = o.grad
    close(d[:, :A].reshape(38, 63, A).permute(2, 0, 1), g12['d_rpn_cls'][0], 1e-5, 1e-8)
    dreg = d[:, A:5 * A].reshape(38, 63, 4 * A).permute(2, 0, 1)
    assert abs(float(dreg.double().abs().sum()) - float(g12['d_rpn_reg_abs'])) <= 1e-5 * float(g12['d_rpn_reg_abs'])
    close(dreg[dreg != 0], g12['d_rpn_reg_nz'], 1e-5, 1e-8)
    assert bool((d[:, 5 * A:] == 0).all())


def test_rcnn_sampling_targets_and_ohem_loss_match_reference(g12):
    tc = C.target_case()
    gt_b, gt_l, props = tc['gt_bboxes'].to(DEV), tc['gt_labels'].to(DEV), tc['proposals'].to(DEV)
    cfg = _cfg(C.RCNN_TRAIN_CFG)
    k, n = gt_b.shape[0], props.shape[0]
    keys = _replay_keys(k + n, np.concatenate([g12['rcnn_pos_inds'], g12['rcnn_neg_inds']]))
    assign_result, res = T.assign_and_sample(props, gt_b, None, gt_l, cfg, keys=keys)
    assert _eq(res.pos_inds, g12['rcnn_pos_inds']) and _eq(res.neg_inds, g12['rcnn_neg_inds'])
    assert _eq(res.bboxes, g12['rcnn_rois'])
    assert _eq(res.pos_is_gt, (g12['rcnn_pos_inds'] < k).astype(np.uint8))
    labels, lw, bt, bw = T.bbox_target([res], [gt_b], [gt_l], cfg, target_stds=(0.1, 0.1, 0.2, 0.2))
    assert _eq(labels, g12['rcnn_labels']) and _eq(lw, g12['rcnn_label_weights']) and _eq(bw, g12['rcnn_bbox_weights'])
    close(bt, g12['rcnn_bbox_targets'], 1e-5, 1e-5)
    # OHEM on fixed head outputs (fused logit layout: 31 class logits | 4 deltas | pad)
    R = labels.shape[0]
    logits = torch.zeros((R, 36))
    logits[:, :31] = tc['cls_score'][:R]
    logits[:, 31:35] = tc['bbox_pred'][:R]
    logits = logits.to(DEV).requires_grad_(True)
    row_loss = native.ce_rows(logits.detach(), 0, 31, labels)
    close(row_loss, g12['ohem_row_loss'], 1e-5, 1e-6)
    post = T.build_sampler([cfg['sampler'], dict(type='OHEMHNLSampler', **C.RCNN_TRAIN_CFG['ohem'])], context=None)[1]
    lw2, bw2, opos, oneg = post.get_ohem_weights(labels, lw.clone(), bw.clone(), row_loss)
    assert _eq(opos, g12['ohem_pos_inds']) and _eq(oneg, g12['ohem_neg_inds'])
    _, sel_counts = post.select(labels, row_loss)
    losses = TO.det_loss_sampled(logits, 0, 31, 31, labels, lw2, bt, bw2, sel_counts, beta=1.0)
    close(losses['loss_cls'], g12['ohem_loss_cls'], 1e-5, 1e-6)
    close(losses['loss_bbox'], g12['ohem_loss_bbox'], 1e-5, 1e-6)
    close(losses['acc'], g12['ohem_acc'], 1e-5, 1e-5)
    losses['total'].sum().backward()
    close(logits.grad[:, :31], g12['ohem_d_cls'], 1e-5, 1e-8)
    close(logits.grad[:, 31:35], g12['ohem_d_reg'], 1e-5, 1e-8)
    assert bool((logits.grad[:, 35] == 0).all())


def test_unsupported_target_options_fail_loudly():
    with pytest.raises(NotImplementedError):
        T.MaxIoUAssigner(0.5, 0.5, gt_max_assign_all=False)
    with pytest.raises(NotImplementedError):
        T.build_sampler(dict(type='IoUBalancedNegSampler', num=10, pos_fraction=0.5))
    asg = T.MaxIoUAssigner(0.5, 0.5, ignore_iof_thr=0.5)
    b = torch.rand(4, 4, device=DEV)
    with pytest.raises(NotImplementedError):
        asg.assign(b, b, gt_bboxes_ignore=b)
    with pytest.raises(native.HvrError):   # more ground-truth boxes than the kernel's LDS table
        native.max_iou_assign(torch.rand(8, 4, device=DEV), torch.rand(300, 4, device=DEV), 0.5, 0.5, 0.5)


def test_hard_proposal_mining_matches_reference_and_oracle(O):
    """HRNMPBBoxHead.hardest_proposal_mining on the device against G13 (the reference's own method) and, for the top-2 of the
    background rows and larger random cases with ties, against the oracle."""
    g13 = np.load(os.path.join(ROOT, 'tests', 'golden', 'g13_mining.npz'))
    labels, all_labels, aff = C.mining_case()
    head = hvrnet_amd.HRNMPBBoxHead(sampler_num=8, t_dim=3, imgs_per_video=3, in_channels=256, num_classes=31, reg_class_agnostic=True)
    anchors, pos, neg = head.hardest_proposal_mining(labels.to(DEV), all_labels.to(DEV), aff[None].to(DEV), None)
    assert _eq(anchors, g13['anchor_idx']) and _eq(pos, g13['hardest_pos_idx']) and _eq(neg, g13['hardest_neg_idx'])
    picks = native.mining_argreduce(aff.to(DEV), labels.to(DEV), all_labels.to(DEV))
    _, _, _, bg2 = O.hardest_proposal_mining(labels, all_labels, aff)
    assert _eq(picks[:, 2:], bg2.numpy())
    # coarse affinities (many exact ties -> the lower index), ragged sizes, a padded row stride
    gen = torch.Generator().manual_seed(5)
    for mq, mk in ((1, 1), (5, 2), (300, 900), (257, 4500)):
        lab = torch.randint(0, 4, (mq,), generator=gen)
        alab = torch.randint(0, 4, (mk,), generator=gen)
        a = (torch.randn((mq, mk), generator=gen) * 4).round() / 4
        buf = torch.zeros((mq, mk + 3))
        buf[:, :mk] = a
        got = native.mining_argreduce(buf.to(DEV)[:, :mk], lab.to(DEV), alab.to(DEV)).cpu()
        diff = alab[None, :] != lab[:, None]
        hi, lo = a.masked_fill(~diff, float('-inf')), a.masked_fill(diff, float('inf'))
        first_max = (hi == hi.max(1, keepdim=True)[0]).float().argmax(1)      # lowest index among the maxima
        first_min = (lo == lo.min(1, keepdim=True)[0]).float().argmax(1)
        assert torch.equal(got[:, 0], first_max) and torch.equal(got[:, 1], first_min), (mq, mk)
        if mk > 1:
            hi2 = hi.clone()
            hi2[torch.arange(mq), first_max] = float('-inf')
            second = (hi2 == hi2.max(1, keepdim=True)[0]).float().argmax(1)
            ok = torch.isfinite(hi2.max(1)[0]) | ~torch.isfinite(hi.max(1)[0])
            # rows with exactly one candidate: the second pick is the lowest masked index other than the first
            assert torch.equal(got[ok, 3], second[ok]), (mq, mk)


@pytest.mark.parametrize('dtype,margin', [(torch.float32, 10.0), (torch.float32, 0.5), (torch.bfloat16, 10.0)])
def test_triplet_margin_standin_matches_its_oracle(O, dtype, margin):
    """The documented stand-in for the un-vendored TripletNonLocalLoss (parity with the reference is unpinned: the library is not
    in its tree): forward value, active-triple count and both gradients against f64 autograd over the oracle's restatement of
    the published TripletMarginLoss; repeated positives / negatives (gradient rows that collide) and inactive triples included."""
    g = torch.Generator().manual_seed(int(margin * 10))
    Mq, Mk, D, n = 40, 70, 1024, 33
    q = (torch.randn((Mq, D), generator=g) * 0.3).to(dtype)
    k = (torch.randn((Mk, D), generator=g) * 0.3).to(dtype)
    a = torch.randint(0, Mq, (n,), generator=g)
    p = torch.randint(0, 6, (n,), generator=g)          # few distinct rows: collisions
    m = torch.randint(0, Mk, (n,), generator=g)
    k[m[:5]] = k[m[:5]] * 6                             # far negatives: inactive triples when the margin is small
    qd, kd = q.double().requires_grad_(True), k.double().requires_grad_(True)
    want, want_active = O.triplet_margin_standin(qd, kd, a, p, m, margin)
    want.backward()
    qg, kg = q.to(DEV).requires_grad_(True), k.to(DEV).requires_grad_(True)
    loss, active = TO.triplet_margin(qg, kg, a.to(DEV), p.to(DEV), m.to(DEV), margin)
    loss.backward()
    assert int(active) == int(want_active) and (margin > 1 or int(active) < n)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    close(loss, want.detach(), tol, tol)
    scale = float(qd.grad.abs().max())
    close(qg.grad, qd.grad, tol * 10, tol * scale)
    close(kg.grad, kd.grad, tol * 10, tol * scale)
    assert qg.grad.dtype == dtype


@pytest.mark.parametrize('hw', [(720, 1280), (360, 480), (600, 1000), (1080, 1920), (37, 53), (480, 270)])
@pytest.mark.parametrize('to_rgb', [False, True])
def test_frame_ingest_matches_the_oracle_bit_for_bit(O, hw, to_rgb):
    """hvr_ingest_frame (resize + normalise + pad in one kernel) against the oracle's restatement of the reference's test
    pipeline: integer resize arithmetic and an exact f32 subtraction -> identical bits; img_meta identical too."""
    from hvrnet_amd.pipelines import FrameIngest
    rng = np.random.RandomState(hw[0])
    frame = rng.randint(0, 256, hw + (3,)).astype(np.uint8)
    std = (1.0, 1.0, 1.0) if not to_rgb else (58.395, 57.12, 57.375)
    want, meta = O.ingest_frame(frame, to_rgb=to_rgb, std=std)
    got = FrameIngest(to_rgb=to_rgb, std=std, device=DEV)(frame)
    assert torch.equal(got['img'].cpu(), want)
    for k in ('ori_shape', 'img_shape', 'pad_shape', 'scale_factor', 'flip'):
        assert got['img_meta'][k] == meta[k], k
    again = FrameIngest(to_rgb=to_rgb, std=std, device=DEV)(torch.from_numpy(frame).to(DEV))    # frame already on the device
    assert torch.equal(again['img'], got['img'])
