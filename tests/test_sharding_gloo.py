"""The N > 1 path (frame sharding + the single all_gather of keypoints) with two gloo processes on CPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rcvpose_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fake_keypoints(frame):
    """Stands in for the per-frame result of the GPU pipeline: deterministic in the global frame index."""
    g = np.random.default_rng(1000 + frame)
    return g.normal(size=(3, 3)) * 100


def _worker(rank, world, port, n_frames, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.frames_of_rank(n_frames, rank, world)
    per = sharding.padded_count(n_frames, world)
    local = torch.zeros((per, 3, 3), dtype=torch.float64)
    for slot, f in enumerate(mine):
        local[slot] = torch.from_numpy(_fake_keypoints(f))
    full = sharding.gather_keypoints(local, n_frames)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [8, 7, 1])
def test_two_rank_gather_equals_single_rank(tmp_path, n_frames):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_frames, str(tmp_path)), nprocs=world, join=True)
    expect = np.stack([_fake_keypoints(f) for f in range(n_frames)])
    for r in range(world):
        got = np.load(tmp_path / ("rank%d.npy" % r))
        assert got.shape == (n_frames, 3, 3)
        assert np.array_equal(got, expect)  # bit-for-bit equal to the world=1 result


def test_single_process_passthrough():
    assert sharding.frames_of_rank(10, 1, 4) == [1, 5, 9]
    assert sharding.padded_count(10, 4) == 3
    k = torch.arange(5 * 9, dtype=torch.float64).reshape(5, 3, 3)
    assert torch.equal(sharding.gather_keypoints(k, 5), k)
