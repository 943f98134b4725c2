"""`python bench.py --gpus N` must start its own ranks: the launcher, the torchrun rendezvous on
127.0.0.1 and the per-step gather are exercised here with two gloo ranks and no GPU work (--selftest-launcher)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-launcher", "--steps", "3", "--frames-per-step", "5"] + extra,
                       capture_output=True, text=True, timeout=600, env=e)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


def test_plain_invocation_with_two_gpus_launches_two_ranks():
    p, lines = _run(["--gpus", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1                      # only rank 0 prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["gather_equals_world1"] is True


def test_single_rank_needs_no_launcher():
    p, lines = _run(["--gpus", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(lines[0])["n_gpus"] == 1


def test_mismatched_world_size_is_refused():
    p, _ = _run(["--gpus", "2"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0
