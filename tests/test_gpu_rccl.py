"""The one exchange of the multi-GPU path (an all_gather of 72 B per frame) through RCCL on real hardware.
A one-GPU box cannot host two ranks, so this is the one-rank group: it still loads RCCL, builds the communicator on the
device and runs the collective on a side stream next to the voting kernels' results, which is everything the N-rank
path does on each rank.  (World size 2 is covered with gloo in tests/test_sharding_gloo.py.)"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from rcvpose_amd import sharding
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
side = torch.cuda.Stream()
g = torch.Generator().manual_seed(5)
for n in (1, 7, 72):
    k = torch.randn((n, 3, 3), generator=g, dtype=torch.float64).cuda()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out = sharding.gather_keypoints(k * 2.0, n, always_collective=True)   # producer kernel + collective on one stream
        host = torch.empty((n, 3, 3), dtype=torch.float64).pin_memory()
        host.copy_(out, non_blocking=True)
    side.synchronize()
    assert out.shape == (n, 3, 3) and torch.equal(host, (k * 2.0).cpu())
    assert sharding.gather_keypoints(k, n) is not None
dist.barrier()
dist.destroy_process_group()
print("rccl-one-rank-ok")
"""


def test_rccl_all_gather_in_a_one_rank_group(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl-one-rank-ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_bench_runs_its_n_rank_path_in_a_one_rank_group():
    """bench.py --rccl-one-rank: communicator on the device, the per-step all_gather on the voting stream, barrier,
    all_reduce(MAX) of the time and the final broadcast -- the statements every rank of `--gpus N` executes."""
    import json

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rccl-one-rank", "--steps", "2", "--warmup", "1",
                        "--frames-per-step", "4", "--no-extras", "--no-cpu-baseline"],
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["ok"] and line["rccl_ranks"] == 1 and line["n_gpus"] == 1 and line["value"] > 0
