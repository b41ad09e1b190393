"""Multi-GPU inference sharding: whole videos per rank, no data-path collective.

A window only ever mixes frames of ONE video, so the unit of work is a video segment and ranks never
exchange activations.  This mirrors the reference:
  * partition   mmdet/datasets/imagenet_vid_sequence.py:117-158 (`get_indices`): walk the videos in order,
                fill a rank until adding the next video would exceed ceil(total_frames / world_size),
                the last rank takes whatever remains;
  * collection  tools/test.py:546-589 (`collect_selsa_results_cpu`): every rank's per-frame results are brought
                to rank 0 and stitched in global frame order.  The reference pickles to a shared tmpdir behind a
                `dist.barrier()`; here `torch.distributed.all_gather_object` carries the same host-side objects
                (RCCL on the GPU box via backend "nccl", gloo in the CPU tests).
"""
import math


def partition_videos(video_lengths, world_size):
    """-> list (per rank) of video indices, the reference's greedy contiguous fill."""
    total = sum(video_lengths)
    avg = int(math.ceil(total / float(world_size)))
    ranks = [[] for _ in range(world_size)]
    cur, filled = 0, 0
    for vid, n in enumerate(video_lengths):
        if filled + n <= avg:
            filled += n
        else:
            if cur != world_size - 1:
                cur += 1
                filled = 0
            filled += n
        ranks[cur].append(vid)
    return ranks


def run_sharded(video_lengths, run_video, rank, world_size, group=None):
    """Every rank runs `run_video(vid) -> list of per-frame results` on its own videos; rank 0 gets the results of
    all videos in global order (None elsewhere).  `run_video` is the only place that touches a GPU."""
    mine = partition_videos(video_lengths, world_size)[rank]
    local = {vid: run_video(vid) for vid in mine}
    if world_size == 1:
        gathered = [local]
    else:
        import torch.distributed as dist
        gathered = [None] * world_size
        dist.all_gather_object(gathered, local, group=group)
    if rank != 0:
        return None
    merged = {}
    for part in gathered:
        merged.update(part)
    assert sorted(merged.keys()) == list(range(len(video_lengths))), 'a video was dropped or duplicated'
    out = []
    for vid in range(len(video_lengths)):
        assert len(merged[vid]) == video_lengths[vid]
        out.extend(merged[vid])
    return out
