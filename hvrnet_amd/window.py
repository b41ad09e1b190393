"""Sliding-window video inference loop (host logic).

Shape of the reference's `multi_selsa_gpu_test` (tools/test.py:143-306) without its dataset / pickle
plumbing: per incoming frame run the backbone once (`model(backbone_feat=True)`), keep the last T
C4 maps in a deque, and once the deque is full emit one key-frame detection per step with
`model(x=deque, img_meta=metas, forward_feat=True, return_loss=False, rescale=True)`.
  first frame of a video  (flag 0): deque padded with copies until it holds (T+1)/2 entries (:201-212)
  middle frames           (flag 2): append; emit when the deque holds T entries            (:214-250)
  last frame              (flag 1): pad to T-1, then append + emit min(seg_len, (T+1)/2) times (:257-300)
The emitted detection belongs to the deque's centre entry (index (T-1)/2, :238-242).
"""
from collections import deque

FIRST, LAST, MIDDLE = 0, 1, 2


def frame_flags(num_frames):
    """key_frame_flag sequence of one video segment (imagenet_vid_sequence.py semantics: 0 first, 1 last, 2 else)."""
    if num_frames == 1:
        return [FIRST]
    return [FIRST] + [MIDDLE] * (num_frames - 2) + [LAST]


class VideoWindowRunner(object):
    """Feeds frames of ONE video through `model`; yields (frame_offset, result) per emitted key frame.

    cache_frames=False: the reference's loop as it stands -- every emitted window recomputes res5 / RPN / RoIAlign /
    fc_new_1 for all T frames (hnmb_rcnn.py:195-222).  cache_frames=True: those per-frame results are computed once when
    the frame arrives (`model.frame_tensors`) and a window runs only the relation stages and the read-out on the
    T cached entries (`model.forward_feat_frames`); same detections, about a third of the work per output frame."""

    def __init__(self, model, window, rescale=True, cache_frames=False):
        assert window % 2 == 1, 'window = 2 * frame_interval + 1'
        self.model, self.T, self.rescale, self.cache_frames = model, window, rescale, cache_frames
        self.center = (window - 1) // 2
        self._reset()

    _entry = None

    def _reset(self):
        self.feats = deque(maxlen=self.T)
        self.offsets = deque(maxlen=self.T)
        self.metas = deque(maxlen=self.T)
        self.entries = deque(maxlen=self.T)

    def _push(self, feat, offset, meta):
        self.feats.append(feat)
        self.offsets.append(offset)
        self.metas.append(meta)
        self.entries.append(self._entry)  # the arriving frame's cached per-frame tensors (None without cache_frames)

    def _emit(self):
        if self.cache_frames:
            result = self.model.forward_feat_frames(list(self.entries), c4s=list(self.feats), rescale=self.rescale)
        else:
            result = self.model(x=self.feats, img=None, img_meta=list(self.metas), forward_feat=True, return_loss=False,
                                rescale=self.rescale)
        return self.offsets[self.center], result

    def step(self, img, img_meta, flag, frame_offset, seg_len=None):
        """One loader iteration; returns the list of (frame_offset, result) emitted by it."""
        out = []
        feat = self.model(img=img, img_meta=[img_meta], backbone_feat=True)[0]
        self._entry = self.model.frame_tensors(feat, img_meta) if self.cache_frames else None
        if flag == FIRST:
            self._reset()
            while len(self.feats) < (self.T + 1) // 2:
                self._push(feat, frame_offset, img_meta)
        elif flag == MIDDLE:
            self._push(feat, frame_offset, img_meta)
            if len(self.feats) == self.T:
                out.append(self._emit())
        elif flag == LAST:
            while len(self.feats) < self.T - 1:
                self._push(feat, frame_offset, img_meta)
            n_end = (self.T + 1) // 2 if seg_len is None else min(seg_len, (self.T + 1) // 2)
            for _ in range(n_end):
                self._push(feat, frame_offset, img_meta)
                out.append(self._emit())
        else:
            raise ValueError('bad key_frame_flag %r' % (flag,))
        return out

    def run_video(self, frames, metas):
        """frames: iterable of [1,3,H,W] tensors of one video. Returns {frame_offset: result}."""
        frames = list(frames)
        flags = frame_flags(len(frames))
        results = {}
        for i, (img, meta, flag) in enumerate(zip(frames, metas, flags)):
            for off, res in self.step(img, meta, flag, i, seg_len=len(frames)):
                results[off] = res
        if len(frames) == 1:  # a one-frame segment is both first and last
            for off, res in self.step(frames[0], metas[0], LAST, 0, seg_len=1):
                results[off] = res
        return results


def window_frames(num_frames, window):
    """{emitted frame offset: the `window` frame indices its deque holds, oldest first} for a video of `num_frames` frames --
    the loop above run on frame indices instead of tensors (tools/test.py:201-212,214-250,257-300): the first frame is
    repeated until the deque holds (T+1)/2 entries, the last one is repeated while the remaining centres are emitted."""
    class _Ids(object):
        frame_tensors = None

        def __call__(self, **kw):
            return [kw['img']] if kw.get('backbone_feat') else list(kw['x'])

    runner = VideoWindowRunner(_Ids(), window)
    out = {}
    flags = frame_flags(num_frames)
    for i, flag in enumerate(flags):
        for off, ids in runner.step(i, None, flag, i, seg_len=num_frames):
            out[off] = ids
    if num_frames == 1:
        for off, ids in runner.step(0, None, LAST, 0, seg_len=1):
            out[off] = ids
    return out


def window_indices(offset, num_frames, window):
    """The deque content (frame indices, oldest first) when frame `offset` of a `num_frames`-frame video is emitted."""
    return window_frames(num_frames, window)[offset]
