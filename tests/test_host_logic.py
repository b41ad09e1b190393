"""CPU-side checks (-m "not gpu"): the plugin surface, config loading, checkpoint key layout,
BatchNorm folding, the window loop, the C-ABI export table, and the rule that the product never
touches the oracle."""
import ast
import ctypes
import os
import re
import sys

import pytest
import torch

import hvrnet_amd
from hvrnet_amd import backbone, native, registry, synthetic, window
from hvrnet_amd.config import Config, ConfigDict, hvr_config, selsa_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CFG = '/root/reference/configs'


def _build(cfg):
    return registry.build_detector(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))


def test_registry_names_match_the_reference_configs():
    for reg, names in [(registry.DETECTORS, ['SelsaRCNN', 'HNMBRCNN']), (registry.BACKBONES, ['ResNet']),
                       (registry.SHARED_HEADS, ['ResLayer']), (registry.ROI_EXTRACTORS, ['SingleRoIExtractor']),
                       (registry.HEADS, ['RPNHead', 'SelsaBBoxHead', 'HRNMPBBoxHead', 'BBoxHead'])]:
        for n in names:
            assert reg.get(n) is not None, (reg.name, n)
    with pytest.raises(KeyError):
        registry.build_from_cfg(dict(type='NoSuchThing'), registry.HEADS)
    with pytest.raises(KeyError):
        registry.HEADS.register_module(hvrnet_amd.RPNHead)  # duplicate registration is an error, as in the reference


@pytest.mark.parametrize('head', ['selsa', 'hvr'])
def test_builtin_config_builds_and_state_dict_keys_are_the_checkpoint_contract(head):
    cfg = selsa_config() if head == 'selsa' else hvr_config()
    model = _build(cfg)
    sd = synthetic.synth_state_dict(head)
    mine = model.state_dict()
    assert set(mine.keys()) == set(sd.keys())
    for k in sd:
        assert tuple(mine[k].shape) == tuple(sd[k].shape), k
    model.load_state_dict(sd, strict=True)
    assert model.key_dim == 7 and model.bbox_head.t_dim == 15 and model.bbox_head.sampler_num == 300
    assert model.feat_from_shared_head is True
    assert tuple(model.state_dict()['bbox_head.selsa_1.linear_out_1.weight'].shape) == (1024, 1024, 1, 1)


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason='reference tree not present')
@pytest.mark.parametrize('name,builtin', [('faster_rcnn_r101_selsa_c5.py', selsa_config), ('faster_rcnn_r101_hrnmp_c5.py', hvr_config)])
def test_reference_config_files_load_unchanged(name, builtin):
    cfg = Config.fromfile(os.path.join(REF_CFG, name))
    assert cfg.test_cfg.rpn.nms_pre == 6000 and cfg.test_cfg.rcnn.nms.iou_thr == 0.3
    mine = builtin(frame_interval=10).model.to_dict()
    theirs = cfg.model.to_dict()
    assert theirs == mine
    assert cfg.test_cfg.to_dict() == builtin(frame_interval=10).test_cfg.to_dict()
    model = _build(Config(dict(model=cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)))
    assert model.bbox_head.t_dim == 21  # the shipped files: frame_interval = 10
    assert type(model).__name__ == cfg.model.type


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason='reference tree not present')
@pytest.mark.parametrize('name,builtin', [('faster_rcnn_r101_selsa_c5.py', 'selsa_train_config'), ('faster_rcnn_r101_hrnmp_c5.py', 'hvr_train_config')])
def test_reference_train_cfg_builds_the_same_target_objects(name, builtin):
    """The shipped configs' train_cfg sections equal the built-in training configs and build assigners / samplers through
    hvrnet_amd.targets exactly as the reference's build_assigner / build_sampler would (same classes, same arguments)."""
    from hvrnet_amd import config as CFG, targets as T
    cfg = Config.fromfile(os.path.join(REF_CFG, name))
    mine = getattr(CFG, builtin)()
    assert cfg.train_cfg.to_dict() == mine.train_cfg.to_dict() and cfg.model.to_dict() == mine.model.to_dict()
    model = _build(Config(dict(model=cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)))
    assert model.key_dim == 0                                        # train_cfg.rcnn.key_dim
    for part in ('rpn', 'rcnn'):
        asg = T.build_assigner(cfg.train_cfg[part].assigner)
        assert isinstance(asg, T.MaxIoUAssigner) and asg.pos_iou_thr == cfg.train_cfg[part].assigner.pos_iou_thr
    smp = T.build_sampler(cfg.train_cfg.rcnn.sampler, context=model)
    if isinstance(cfg.train_cfg.rcnn.sampler, (list, tuple)):
        assert [type(x).__name__ for x in smp] == ['RandomSampler', 'OHEMHNLSampler'] and (smp[0].num, smp[1].num) == (300, 128)
    else:
        assert type(smp).__name__ == 'RandomSampler' and smp.num == 128 and smp.add_gt_as_proposals
    rpn_smp = T.build_sampler(cfg.train_cfg.rpn.sampler)
    assert (rpn_smp.num, rpn_smp.pos_fraction, rpn_smp.add_gt_as_proposals) == (256, 0.5, False)


def test_configdict_access_patterns_used_by_the_path():
    c = ConfigDict(score_thr=0.001, nms=dict(type='nms', iou_thr=0.3), max_per_img=300)
    assert hasattr(c, 'nms') and not hasattr(c, 'nope') and c.nms.iou_thr == 0.3 and c.get('x', 5) == 5
    d = c.nms.copy()
    d.pop('type')
    assert 'type' in c.nms


def test_fold_conv_bn_equals_conv_then_eval_bn():
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(8, 16, 3, padding=1, bias=False)
    bn = torch.nn.BatchNorm2d(16).eval()
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_()
    bn.running_mean.normal_()
    bn.running_var.uniform_(0.5, 1.5)
    x = torch.randn(2, 8, 9, 11)
    w, b = backbone.fold_conv_bn(conv, bn, torch.float32)
    assert w.shape == (16, 3, 3, 8)
    y = torch.nn.functional.conv2d(x, w.permute(0, 3, 1, 2), b, padding=1)
    torch.testing.assert_close(y, bn(conv(x)), rtol=1e-5, atol=1e-5)


def test_gpu_only_modules_fail_loudly_on_cpu():
    model = _build(hvr_config())
    with pytest.raises(NotImplementedError):
        model.backbone(torch.zeros(1, 3, 64, 64))
    with pytest.raises(NotImplementedError):
        hvrnet_amd.ops.RoIAlign(7, 1 / 16, 2)(torch.zeros(1, 4, 8, 8), torch.zeros(1, 5))
    with pytest.raises(NotImplementedError):
        hvrnet_amd.ops.nms(torch.rand(4, 5), 0.5)
    # the training step has no CPU path either: it dies in the first HIP op, not in a silent torch fallback
    from hvrnet_amd.config import hvr_train_config, selsa_train_config
    meta = dict(img_shape=(64, 64, 3), pad_shape=(64, 64, 3), scale_factor=1.0, flip=False)
    for cfg, frames in ((hvr_train_config(nms_post=8, rcnn_sampler_num=4), 15), (selsa_train_config(nms_post=8, rcnn_sampler_num=4), 3)):
        trainable = hvrnet_amd.enable_training(_build(cfg))
        with pytest.raises(NotImplementedError):
            trainable(torch.zeros(frames, 3, 64, 64), [meta] * frames, return_loss=True,
                      gt_bboxes=[torch.tensor([[4., 4., 40., 40.]])] * frames, gt_labels=[torch.tensor([3])] * frames)
    with pytest.raises(ValueError):       # a detector built without train_cfg says so
        model(torch.zeros(15, 3, 64, 64), [meta] * 15, return_loss=True, gt_bboxes=[None] * 15, gt_labels=[None] * 15)


class _FakeModel(object):
    def __init__(self):
        self.windows = []

    def __call__(self, img=None, img_meta=None, backbone_feat=False, forward_feat=False, x=None, **kw):
        if backbone_feat:
            return (img,)
        self.windows.append(list(x))
        return list(x)


@pytest.mark.parametrize('n_frames,T', [(40, 15), (9, 15), (1, 5), (6, 5)])
def test_window_loop_matches_reference_shape(n_frames, T):
    """Every frame gets exactly one detection; windows are the frame's +-T//2 neighbours with edge replication."""
    m = _FakeModel()
    res = window.VideoWindowRunner(m, T).run_video(list(range(n_frames)), [dict(i=i) for i in range(n_frames)])
    assert sorted(res.keys()) == list(range(n_frames))
    half = T // 2
    for f, win in res.items():
        assert len(win) == T
        want = [min(max(f + d, 0), n_frames - 1) for d in range(-half, half + 1)]
        assert win == want, (f, win)


def test_c_abi_exports_every_declared_symbol():
    header = open(native.HEADER_PATH).read()
    declared = set(re.findall(r'\b(hvr_[a-z0-9_]+)\s*\(', header))
    declared -= {'hvr_gemm_desc', 'hvr_conv_desc', 'hvr_rpn_desc'}
    assert declared == set(native.SYMBOLS.keys()), declared ^ set(native.SYMBOLS.keys())
    if not os.path.exists(native.LIB_PATH):
        native.build()
    lib = ctypes.CDLL(native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert native.lib().hvr_abi_version() == native.ABI_VERSION == 6


def test_bottleneck_expand_convs_take_the_panel_kernel():
    """hvr_conv2d_path (no launch, no GPU): the channel-expanding 1x1 + residual convs of layers 1-3 (resnet.py:248-264) go to
    expand.hip at every batch size the window uses (15 frames, the 8 / 7-frame groups, one cached frame), and so do res5's
    (K = 512); everything else -- convs without a residual, 3x3 convs, the f32 parity mode, fewer than 128 pixels -- stays on the tile
    engine; tile hint 13 forces the panel kernel where it applies (no residual needed) and is ignored where it does not."""
    if not os.path.exists(native.LIB_PATH):
        native.build()
    layers = [(152, 252, 64), (76, 126, 128), (38, 63, 256)]
    for frames in (15, 8, 7, 1):
        for H, W, c in layers:
            # from K = 256 on, a batch whose 288 x 256 tile grid covers the chip takes the big-tile kernel (bigtile.hip, path 3)
            want = 3 if (c == 256 and frames > 1) else 1
            assert native.conv2d_path(frames, H, W, c, 4 * c) == want, (frames, c)
    assert native.conv2d_path(15, 38, 63, 512, 2048) == 3                       # res5's expand convs: big tiles (117 vs 138 us)
    assert native.conv2d_path(1, 38, 63, 512, 2048) == 1                        # one frame: the panel kernel (8 waves x 16 rows)
    assert native.conv2d_path(15, 38, 63, 512, 2048, tile=16) == 1              # a throughput caller keeps the panel kernel here
    assert native.conv2d_path(15, 76, 126, 128, 512, resid=False) == 0          # the panel kernel's automatic choice wants the residual
    assert native.conv2d_path(15, 38, 63, 256, 1024, resid=False, tile=13) == 1
    assert native.conv2d_path(15, 38, 63, 1024, 4096, tile=13) == 0             # K = 1024: not a shape the kernel has
    assert native.conv2d_path(15, 38, 63, 1024, 256, resid=False) == 0          # the reducing 1x1: 125 big tiles are half a chip ...
    assert native.conv2d_path(15, 38, 63, 1024, 256, resid=False, tile=16) == 3 # ... which a throughput caller takes (four graph lanes: +0.4 ... 0.8 %, profiles/r04_lanes.txt)
    assert native.conv2d_path(15, 38, 63, 256, 256, k=3, pad=1, resid=False) == 0
    assert native.conv2d_path(15, 38, 63, 512, 512, k=3, pad=2, dil=2, resid=False) == 3   # res5's 3x3: 250 big tiles
    # layer 1's conv2 (3x3, 64 -> 64, no residual) has its own persistent kernel (conv3x3.hip); nothing else does
    for frames in (15, 8, 7, 1):
        assert native.conv2d_path(frames, 152, 252, 64, 64, k=3, pad=1, resid=False) == 2
    assert native.conv2d_path(15, 152, 252, 64, 64, k=3, pad=1, resid=False, tile=1) == 0
    assert native.conv2d_path(15, 152, 252, 64, 64, k=3, pad=1, resid=True) == 0
    assert native.conv2d_path(15, 76, 126, 128, 128, k=3, pad=1, resid=False) == 0
    assert native.conv2d_path(15, 152, 252, 64, 64, k=3, pad=2, dil=2, resid=False) == 0
    assert native.conv2d_path(1, 19, 23, 64, 64, k=3, pad=1, resid=False) == 0           # fewer than four tiles
    assert native.conv2d_path(15, 38, 63, 256, 1024, dtype=torch.float32) == 0
    assert native.conv2d_path(15, 38, 63, 256, 1024, out_f32=True) == 0
    assert native.conv2d_path(1, 9, 14, 64, 256) == 0                           # 126 pixels < one 128-row panel
    assert native.conv2d_path(1, 8, 16, 64, 256) == 1
    assert native.conv2d_path(15, 38, 63, 256, 1024 + 32) == 0                  # not whole 64-channel chunks
    assert native.conv2d_path(15, 38, 63, 48, 192) < 0                          # Cin not a K-step multiple: rejected


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'hvrnet_amd')):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith('.py'):
                tree = ast.parse(open(path).read())
                for node in ast.walk(tree):
                    mods = []
                    if isinstance(node, ast.Import):
                        mods = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom):
                        mods = [node.module or '']
                    bad += [(path, m) for m in mods if m.split('.')[0] == 'oracle']
            elif f.endswith(('.hip', '.h', '.cpp', '.sh')):
                if 'oracle' in open(path).read():
                    bad.append((path, 'mentions oracle'))
    assert not bad, bad


# ---- the build's assembly lint (hvrnet_amd/csrc/check_asm_waits.py): what it must catch, what it must let through ----
def _asm_kernel(body):
    return ['_Z4kernv:'] + body + ['\ts_endpgm']


def test_asm_wait_lint_flags_a_loop_header_copy_of_an_unlanded_fragment():
    """Round 3's race, in miniature: a fragment set renamed across iterations -- the compiler's copy at the loop header reads the
    register an inline-asm ds_read of the previous iteration wrote, in front of the lgkmcnt wait."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('check_asm_waits', os.path.join(ROOT, 'hvrnet_amd', 'csrc', 'check_asm_waits.py'))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    loop = lambda header: _asm_kernel([                                                             # noqa: E731
        '.LBB0_1:'] + header + [
        '\t;;#ASMSTART', '\ts_waitcnt lgkmcnt(0)', '\t;;#ASMEND',
        '\tv_mfma_f32_16x16x32_bf16 v[20:23], v[10:13], v[0:3], v[20:23]',
        '\t;;#ASMSTART', '\tds_read_b128 v[30:33], v40 offset:0', '\t;;#ASMEND',
        '\tv_mfma_f32_16x16x32_bf16 v[24:27], v[10:13], v[4:7], v[24:27]',
        '\ts_cbranch_scc1 .LBB0_1'])
    bad = lint.check_kernel('k', loop(['\tv_mov_b32_e32 v10, v30', '\tv_mov_b32_e32 v11, v31']))
    assert bad and 'v_mov_b32_e32 v10, v30' in bad[0], bad
    # the same loop with the set used in place (no rename, no copy) passes, and so does a copy placed behind the wait
    assert lint.check_kernel('k', loop([])) == []
    ok = _asm_kernel(['.LBB0_1:', '\t;;#ASMSTART', '\ts_waitcnt lgkmcnt(0)', '\t;;#ASMEND', '\tv_mov_b32_e32 v10, v30',
                      '\t;;#ASMSTART', '\tds_read_b128 v[30:33], v40 offset:0', '\t;;#ASMEND', '\ts_cbranch_scc1 .LBB0_1'])
    assert lint.check_kernel('k', ok) == []
    # counted waits: lgkmcnt(1) covers every read but the youngest
    part = _asm_kernel(['.LBB0_1:',
                        '\t;;#ASMSTART', '\tds_read_b128 v[30:33], v40 offset:0', '\t;;#ASMEND',
                        '\t;;#ASMSTART', '\tds_read_b128 v[34:37], v40 offset:2048', '\t;;#ASMEND',
                        '\t;;#ASMSTART', '\ts_waitcnt lgkmcnt(1)', '\t;;#ASMEND',
                        '\tv_mfma_f32_16x16x32_bf16 v[20:23], v[30:33], v[0:3], v[20:23]',
                        '\tv_mfma_f32_16x16x32_bf16 v[24:27], v[34:37], v[4:7], v[24:27]',
                        '\t;;#ASMSTART', '\ts_waitcnt lgkmcnt(0)', '\t;;#ASMEND', '\ts_cbranch_scc1 .LBB0_1'])
    bad = lint.check_kernel('k', part)
    assert len(bad) == 1 and 'v[34:37]' in bad[0], bad
    # an untracked global load: a compiler-made copy of its destination in front of the vmcnt wait is flagged, a computation is the
    # author's business (it sits behind `landed()`, which the scan cannot see through correlated branches)
    vm = _asm_kernel(['.LBB0_1:', '\t;;#ASMSTART', '\tglobal_load_dword v50, v[60:61], off', '\t;;#ASMEND', '\tv_mov_b32_e32 v51, v50',
                      '\ts_waitcnt vmcnt(0)', '\ts_cbranch_scc1 .LBB0_1'])
    assert lint.check_kernel('k', vm)


def test_asm_wait_lint_passes_on_the_built_library():
    """When this checkout has been built (build.sh keeps the device assembly of the kernels with hand-counted waits), the lint that
    build.sh ran must still pass on those files -- i.e. the library next to them was not linked from unchecked objects."""
    import glob
    import subprocess
    files = sorted(glob.glob(os.path.join(ROOT, 'hvrnet_amd', 'csrc', 'build', '*-hip-amdgcn-amd-amdhsa-gfx950.s')))
    files = [f for f in files if not os.path.basename(f).startswith('nms')]
    if not files:
        pytest.skip('no device assembly here (the library was built elsewhere)')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'hvrnet_amd', 'csrc', 'check_asm_waits.py')] + files, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_the_tolerance_constants_are_frozen():
    """hvrnet_amd/parity.py is frozen as the round-5 judge accepted it ("Next round" 1a): class indices exact, scores within
    1e-3, boxes within 1e-3 px + 1.3e-6 x extent; round 4's fixed bar and north_star's literal 1e-3 px are reported beside it."""
    from hvrnet_amd import parity
    assert (parity.TOL_SCORE, parity.TOL_BOX_PX, parity.BOX_RTOL, parity.NMS_TIE_BAND) == (1e-3, 1e-3, 1.3e-6, 1e-4)
    assert abs(parity.box_bar(1000.0) - 2.3e-3) < 1e-12
    st = dict(class_flips=0, max_score_err=5e-4, max_box_err=1.04e-3, max_box_excess=1.04e-3 - 1.3e-3)
    assert parity.within_tolerance(st) and parity.fixed_bar_r04(st) and not parity.literal_1e3(st)
    assert parity.literal_1e3(dict(st, max_box_err=9.9e-4))
    assert not parity.within_tolerance(dict(st, class_flips=1))


def test_nms_threshold_ties_names_the_pair_at_issue():
    """parity.nms_threshold_ties: two per-frame proposal lists that differ because one side kept a box the other suppressed -> the
    (box, suppressor) pair whose IoU is nearest the threshold, per differing frame; only a pair inside the band is a tie."""
    import numpy as np
    from hvrnet_amd import parity
    keep = np.array([[0, 0, 99, 99, 0.9], [200, 200, 300, 300, 0.7]], dtype=np.float64)
    # a 100 x h box over a 100 x 100 box: IoU = 100 / h  ->  h = 100 / 0.7 puts the pair on the threshold
    h_tie, h_far = 100.0 / 0.70000005, 100.0 / 0.705
    tail = np.array([[400, 400, 500, 500, 0.6]], dtype=np.float64)
    for h, is_tie in ((h_tie, True), (h_far, False)):
        extra = np.array([[0, 0, 99, h - 1, 0.8]], dtype=np.float64)
        got = np.concatenate([keep[:1], extra, keep[1:]], 0)       # this side kept the box ...
        want = np.concatenate([keep, tail], 0)                     # ... the other suppressed it and its 300-th survivor moved up
        assert parity.proposal_lists_equal([got, keep], [want, keep]) == [False, True]
        ties = parity.nms_threshold_ties([got, keep], [want, keep], thr=0.7)
        assert len(ties) == 1 and ties[0]['frame'] == 0 and ties[0]['side'] == 'got'
        assert ties[0]['is_tie'] is is_tie and abs(ties[0]['iou'] - 100.0 / h) < 1e-9
        assert ties[0]['suppressor'][:4] == [0, 0, 99, 99]
    assert parity.nms_threshold_ties([keep], [keep]) == []


def test_training_target_bookkeeping_is_lazy_and_equal_to_the_eager_form():
    """Round 6 host-side changes of the training step (no kernel involved, CPU tensors): AssignResult.add_gt_ defers the max_overlaps cat,
    SamplingResult builds gt_flags on first use, gathers `bboxes` once and reads several results' counts in ONE host copy
    (SamplingResult.resolve); detectors._sampled_rois == the per-frame new_full / cat form of bbox2roi (transforms.py:114-136)."""
    from hvrnet_amd import detectors, targets as T
    g = torch.Generator().manual_seed(3)
    ar = T.AssignResult(2, torch.tensor([0, 1, 2, -1, 0]), torch.tensor([0.1, 0.8, 0.9, -1.0, 0.2]), torch.tensor([0, 5, 7, 0, 0]))
    ar.add_gt_(torch.tensor([5, 7]))
    assert ar.gt_inds.tolist() == [1, 2, 0, 1, 2, -1, 0] and ar.labels.tolist() == [5, 7, 0, 5, 7, 0, 0]
    assert torch.equal(ar.max_overlaps, torch.tensor([1.0, 1.0, 0.1, 0.8, 0.9, -1.0, 0.2]))
    results, want = [], []
    for i, (np_, nn_) in enumerate([(2, 3), (1, 4), (3, 2)]):
        boxes = torch.rand((9, 4), generator=g) * 100
        inds = torch.randperm(9, generator=g)
        r = T.SamplingResult(inds, torch.tensor([np_, nn_], dtype=torch.int32), boxes, boxes[:2], ar, 2)
        results.append(r)
        b = boxes[inds[:np_ + nn_]]
        want.append(torch.cat([b.new_full((b.shape[0], 1), float(i)), b], 1))
    assert all(r._n is None for r in results)
    T.SamplingResult.resolve(results)
    assert [r._n for r in results] == [(2, 3), (1, 4), (3, 2)]
    assert results[0].bboxes is results[0].bboxes                       # gathered once
    assert results[0].gt_flags.tolist() == [1, 1, 0, 0, 0, 0, 0, 0, 0] and results[0].pos_inds.numel() == 2 and results[0].neg_inds.numel() == 3
    rois, n = detectors._sampled_rois(results)
    assert n == [5, 5, 5] and torch.equal(rois, torch.cat(want, 0))     # equal counts: the repeat_interleave form
    results[1]._n, results[1]._bboxes = (1, 2), None
    rois2, n2 = detectors._sampled_rois(results)
    assert n2 == [5, 3, 5] and torch.equal(rois2[:5], want[0]) and torch.equal(rois2[5:8], want[1][:3]) and torch.equal(rois2[8:], want[2])


def test_loss_backward_scales_columns_without_a_host_read():
    """train_ops._scale_columns: the fused logits' gradient under an upstream gradient g [2] of (loss_cls, loss_bbox): class columns x g[0],
    every other column x g[1] -- what the unit-gradient check (a device read-back in the middle of the backward pass) became in round 6."""
    from hvrnet_amd import train_ops as TO
    d = torch.arange(3 * 36, dtype=torch.float32).reshape(3, 36)
    gsc = torch.tensor([2.0, -0.5])
    out = TO._scale_columns(d, gsc, 0, 31)
    assert torch.equal(out[:, :31], d[:, :31] * 2.0) and torch.equal(out[:, 31:], d[:, 31:] * -0.5)
    assert torch.equal(TO._scale_columns(d, torch.ones(2), 0, 31), d)  # the usual (loss_cls + loss_bbox).backward(): unchanged bit for bit
    o = torch.arange(2 * 16, dtype=torch.float32).reshape(2, 16)
    out = TO._scale_columns(o, gsc, 0, 3)                               # RPN: A = 3 objectness columns, then 4 A deltas (+ padding)
    assert torch.equal(out[:, :3], o[:, :3] * 2.0) and torch.equal(out[:, 3:], o[:, 3:] * -0.5)
