"""hvrnet_amd.evaluation (host numpy: the tools/vid_eval.py path of the reference's eval_map) against G14 = the reference's own
mean_ap.eval_map on the same detections / annotations: mAP, per-class AP, counts and the full recall / precision curves of one
class must agree bit for bit (same float32 arithmetic, same argsort calls), in all four modes; plus the hand-checkable cases."""
import os

import numpy as np
import pytest

from hvrnet_amd import evaluation as E
from tests.golden import cases as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def g14():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'g14_eval_map.npz'))


@pytest.mark.parametrize('tag', ['plain', 'ignore', 'thr75', 'scales'])
def test_eval_map_matches_the_reference(g14, tag):
    dets, gtb, gtl, gti = C.eval_case()
    kw = dict(plain=dict(), ignore=dict(gt_ignore=gti), thr75=dict(iou_thr=0.75),
              scales=dict(gt_ignore=gti, scale_ranges=[(0, 64), (64, 128), (128, 1e5)]))[tag]
    m, res = E.eval_map(dets, gtb, gtl, dataset=tuple('c%d' % i for i in range(len(dets[0]))), print_summary=False, **kw)
    assert np.array_equal(np.asarray(m, dtype=np.float64), g14[tag + '_map'])
    assert np.array_equal(np.stack([np.atleast_1d(r['ap']) for r in res]), g14[tag + '_ap'])
    assert np.array_equal(np.stack([np.atleast_1d(r['num_gts']) for r in res]), g14[tag + '_num_gts'])
    assert [r['num_dets'] for r in res] == g14[tag + '_num_dets'].tolist()
    assert np.array_equal(np.asarray(res[0]['recall']), g14[tag + '_recall_c0'])
    assert np.array_equal(np.asarray(res[0]['precision']), g14[tag + '_precision_c0'])


def test_tpfp_hand_cases():
    gt = np.array([[0, 0, 9, 9], [20, 20, 29, 29]], dtype=np.float32)
    det = np.array([[0, 0, 9, 9, 0.9],      # exact hit on box 0 -> tp
                    [0, 0, 9, 8, 0.8],      # second claim on box 0 (IoU 0.9) -> fp
                    [20, 20, 29, 24, 0.7],  # IoU exactly 0.5 with box 1 -> tp (>= threshold)
                    [50, 50, 59, 59, 0.6]], dtype=np.float32)   # no overlap -> fp
    tp, fp = E.tpfp_default(det, gt, np.zeros(2, dtype=bool), 0.5)
    assert tp.tolist() == [[1, 0, 1, 0]] and fp.tolist() == [[0, 1, 0, 1]]
    # an ignored box absorbs its detections: neither tp nor fp, and a later detection cannot "cover" it either
    tp, fp = E.tpfp_default(det, gt, np.array([True, False]), 0.5)
    assert tp.tolist() == [[0, 0, 1, 0]] and fp.tolist() == [[0, 0, 0, 1]]
    # no ground truth: all false positives; no detections: empty
    tp, fp = E.tpfp_default(det, gt[:0], np.zeros(0, dtype=bool), 0.5)
    assert tp.sum() == 0 and fp.tolist() == [[1, 1, 1, 1]]
    tp, fp = E.tpfp_default(det[:0], gt, np.zeros(2, dtype=bool), 0.5)
    assert tp.shape == (1, 0) and fp.shape == (1, 0)


def test_average_precision_known_answers():
    # perfect ranking: AP 1; one miss at the end: recall stops at 0.5 -> AP 0.5
    assert E.average_precision(np.array([0.5, 1.0]), np.array([1.0, 1.0])) == pytest.approx(1.0)
    assert E.average_precision(np.array([0.5, 0.5]), np.array([1.0, 0.5])) == pytest.approx(0.5)
    assert E.average_precision(np.array([0.0, 0.5, 1.0]), np.array([0.0, 0.5, 2 / 3])) == pytest.approx(2 / 3)
    with pytest.raises(ValueError):
        E.average_precision(np.array([1.0]), np.array([1.0]), mode='nope')


def test_vid_eval_merges_ignore_boxes_like_the_reference_script():
    dets, gtb, gtl, gti = C.eval_case(n_img=6)
    anns = []
    for b, l, ig in zip(gtb, gtl, gti):
        anns.append(dict(bboxes=b[~ig], labels=l[~ig], bboxes_ignore=b[ig], labels_ignore=l[ig]))
    m, res = E.vid_eval([d + [np.zeros((0, 5), np.float32)] * 25 for d in dets], anns, print_summary=False)
    order = [np.concatenate([np.nonzero(~ig)[0], np.nonzero(ig)[0]]) for ig in gti]
    m2, _ = E.eval_map([d + [np.zeros((0, 5), np.float32)] * 25 for d in dets], [b[o] for b, o in zip(gtb, order)],
                       [l[o] for l, o in zip(gtl, order)], gt_ignore=[ig[o] for ig, o in zip(gti, order)], dataset=E.VID_CLASSES,
                       print_summary=False)
    assert m == m2 and len(res) == 30
