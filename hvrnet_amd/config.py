"""Config loading (the ~50-line stand-in for `mmcv.Config.fromfile`, which is not installed) and
the two hot-path configurations as built-in dicts.

The reference's configs are plain Python files executed for their module-level variables
(configs/faster_rcnn_r101_{selsa,hrnmp}_c5.py); consumers need attribute access, `.get`,
`.copy()`, `hasattr(cfg, 'nms')` and nested attributes such as `test_cfg.bbox_head.t_dim`
(bbox_nms.py:32-33, hnmb_rcnn.py:44-48).  `Config.fromfile` accepts those files unchanged;
`selsa_config()` / `hvr_config()` give the same model/test settings without the reference tree
(with `frame_interval` exposed: the shipped files say 10 -> T=21, BASELINE.json fixes T=15 -> 7).
"""
import os
import types


class ConfigDict(dict):
    """dict with attribute access; nested dicts are wrapped on the way in."""

    def __init__(self, *args, **kwargs):
        super(ConfigDict, self).__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return ConfigDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(ConfigDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super(ConfigDict, self).__setitem__(k, ConfigDict._wrap(v))

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError("'ConfigDict' object has no attribute '%s'" % name)

    def __setattr__(self, name, value):
        self[name] = value

    def copy(self):
        return ConfigDict(dict.copy(self))

    def to_dict(self):
        def plain(v):
            if isinstance(v, ConfigDict):
                return {k: plain(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return type(v)(plain(x) for x in v)
            return v
        return plain(self)


class Config(object):

    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg_dict', ConfigDict(cfg_dict or {}))
        object.__setattr__(self, 'filename', filename)

    @staticmethod
    def fromfile(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise IOError('config file %s does not exist' % filename)
        scope = {'__file__': filename, '__name__': '_hvr_config_'}
        with open(filename, 'r') as f:
            exec(compile(f.read(), filename, 'exec'), scope)
        cfg = {k: v for k, v in scope.items()
               if not k.startswith('__') and not isinstance(v, (types.ModuleType, types.FunctionType, type))}
        return Config(cfg, filename)

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def __contains__(self, key):
        return key in self._cfg_dict


# ----------------------------------------------------------------------------------------------
# built-in equivalents of the two reference configs (model + test_cfg; values cited)
# ----------------------------------------------------------------------------------------------
def _model(net_type, bbox_type, head_extra):
    norm_cfg = dict(type='BN', requires_grad=False)
    return dict(
        type=net_type,
        # configs/faster_rcnn_r101_selsa_c5.py:19-29 == ..._hrnmp_c5.py:40-50
        backbone=dict(type='ResNet', depth=101, num_stages=3, strides=(1, 2, 2), dilations=(1, 1, 1), out_indices=(2,),
                      frozen_stages=1, style='caffe', norm_eval=True, norm_cfg=norm_cfg),
        # :30-39 / :51-60
        shared_head=dict(type='ResLayer', depth=101, stage=3, stride=1, dilation=2, style='caffe', norm_eval=True,
                         norm_cfg=norm_cfg, external_conv=True),
        # :40-51 / :61-72
        rpn_head=dict(type='RPNHead', in_channels=1024, feat_channels=512, anchor_scales=[4, 8, 16, 32],
                      anchor_ratios=[0.5, 1.0, 2.0], anchor_strides=[16], target_means=[.0, .0, .0, .0],
                      target_stds=[1.0, 1.0, 1.0, 1.0],
                      loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                      loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0)),
        # :52-57 / :73-78
        bbox_roi_extractor=dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', out_size=7, sample_num=2),
                                out_channels=1024, featmap_strides=[16], feat_from_shared_head=True),
        # :58-72 / :79-95
        bbox_head=dict(type=bbox_type, with_avg_pool=False, in_channels=256, fc_feat_dim=1024, roi_feat_size=7,
                       num_classes=31, target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2],
                       reg_class_agnostic=True,
                       loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                       loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0), **head_extra))


def _test_cfg(frame_interval, nms_post, test_branches=1):
    # configs/faster_rcnn_r101_hrnmp_c5.py:139-160 (selsa: :124-144)
    return dict(
        rpn=dict(nms_across_levels=False, nms_pre=6000, nms_post=nms_post, max_num=nms_post, nms_thr=0.7, min_bbox_size=0),
        rcnn=dict(score_thr=0.001, nms=dict(type='nms', iou_thr=0.3), max_per_img=300, key_dim=frame_interval),
        bbox_head=dict(sampler_num=nms_post, t_dim=(frame_interval * 2 + 1) * test_branches,
                       key_dim=(frame_interval * 2 + 1) * int((test_branches - 1) / 2) + frame_interval),
        relation_setup=dict(shuffle=False, video_shuffle=True, has_rpn=True, frame_interval=frame_interval, frame_stride=1))


def selsa_config(frame_interval=7, nms_post=300):
    """SelsaRCNN + SelsaBBoxHead (configs/faster_rcnn_r101_selsa_c5.py), test-time settings."""
    model = _model('SelsaRCNN', 'SelsaBBoxHead', dict(sampler_num=128, t_dim=3))
    return Config(dict(model=model, train_cfg=None, test_cfg=_test_cfg(frame_interval, nms_post)))


def _train_cfg(nms_post, rcnn_sampler_num, ohem=True):
    # configs/faster_rcnn_r101_selsa_c5.py:74-122
    first = dict(type='RandomSampler', num=nms_post, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True)
    second = dict(type='OHEMHNLSampler', num=rcnn_sampler_num, pos_fraction=0.25, neg_pos_ub=-1)
    return dict(
        rpn=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, ignore_iof_thr=-1),
                 sampler=dict(type='RandomSampler', num=256, pos_fraction=0.5, neg_pos_ub=-1, add_gt_as_proposals=False),
                 allowed_border=0, pos_weight=-1, debug=False),
        rpn_proposal=dict(nms_across_levels=False, nms_pre=6000, nms_post=nms_post, max_num=nms_post, nms_thr=0.7, min_bbox_size=0),
        rcnn=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, ignore_iof_thr=-1),
                  sampler=[first, second] if ohem else first, key_dim=0, pos_weight=-1, debug=False))


def selsa_train_config(nms_post=300, rcnn_sampler_num=128, t_dim=3, ohem=True):
    """SelsaRCNN with its training settings (configs/faster_rcnn_r101_selsa_c5.py: model + train_cfg, key frame first)."""
    model = _model('SelsaRCNN', 'SelsaBBoxHead', dict(sampler_num=rcnn_sampler_num, t_dim=t_dim))
    return Config(dict(model=model, train_cfg=_train_cfg(nms_post, rcnn_sampler_num, ohem), test_cfg=_test_cfg((t_dim - 1) // 2, nms_post)))


def hvr_train_config(nms_post=300, rcnn_sampler_num=128, imgs_per_video=3, chosen_videos=3):
    """HNMBRCNN with its training settings (configs/faster_rcnn_r101_hrnmp_c5.py:4-5,34-35,79-95,97-137): one RandomSampler of
    `rcnn_sampler_num` per frame, head t_dim = imgs_per_video * chosen_videos."""
    model = _model('HNMBRCNN', 'HRNMPBBoxHead', dict(sampler_num=rcnn_sampler_num, imgs_per_video=imgs_per_video,
                                                     t_dim=imgs_per_video * chosen_videos))
    train = _train_cfg(nms_post, rcnn_sampler_num, ohem=False)
    train['rcnn']['sampler'] = dict(type='RandomSampler', num=rcnn_sampler_num, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True)
    test = _test_cfg(1, nms_post)
    test['bbox_head'].update(key_dim=0)
    return Config(dict(model=model, train_cfg=train, test_cfg=test))


def hvr_config(frame_interval=7, nms_post=300):
    """HNMBRCNN + HRNMPBBoxHead (configs/faster_rcnn_r101_hrnmp_c5.py), test-time settings."""
    model = _model('HNMBRCNN', 'HRNMPBBoxHead', dict(sampler_num=128, imgs_per_video=3, t_dim=9))
    return Config(dict(model=model, train_cfg=None, test_cfg=_test_cfg(frame_interval, nms_post)))
