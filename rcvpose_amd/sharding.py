"""Frame sharding over the GPUs of one node.

The path shards embarrassingly: frames are independent (reference AccumulatorSpace.py:553), so global
frame f goes to rank f % world, every rank holds all three keypoint networks, and nothing crosses GPUs
during compute.  The only exchange is one all_gather of the recovered keypoints (float64 [F_local,3,3],
72 B per frame) per batch -- RCCL over xGMI on the GPU box (torch.distributed backend "nccl"), gloo in the
CPU tests.  The reference has no counterpart (it is single-process; its only parallel construct is
nn.DataParallel for training, train.py:31-34).
"""


def frames_of_rank(n_frames, rank, world):
    """Global frame indices owned by `rank` (round-robin)."""
    return list(range(rank, n_frames, world))


def padded_count(n_frames, world):
    """Frames per rank after padding to equal share (all_gather needs equal shapes)."""
    return (n_frames + world - 1) // world


def gather_keypoints(local_kpts, n_frames, group=None, always_collective=False):
    """local_kpts: [ceil(n_frames/world), G, 3] tensor of this rank's frames (rows past the rank's last
    frame are padding).  Returns the [n_frames, G, 3] tensor in global frame order on every rank.
    always_collective: issue the all_gather even in a one-rank group (tests: the RCCL path on a one-GPU box)."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not always_collective):
        return local_kpts[:n_frames]
    world = dist.get_world_size(group)
    per = padded_count(n_frames, world)
    assert local_kpts.shape[0] == per, (local_kpts.shape, per)
    parts = [torch.empty_like(local_kpts) for _ in range(world)]
    dist.all_gather(parts, local_kpts.contiguous(), group=group)
    stacked = torch.stack(parts, dim=1)  # [per, world, G, 3]: row-major flattening = frame i*world + r
    return stacked.reshape(per * world, *local_kpts.shape[1:])[:n_frames]
