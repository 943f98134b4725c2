#!/usr/bin/env python3
"""cProfile of the frame loop of `main.py --mode=test` on the synthetic LINEMOD class (where the host time of the driver goes)."""
import cProfile, os, pstats, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_driver as T
from rcvpose_amd.main import main as rcv_main
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
with tempfile.TemporaryDirectory() as tmp:
    root = tmp + "/"
    cls, _ = T._write_dataset(root, n_frames=n)
    T._write_lm_checkpoints(root, cls, os.path.join(ROOT, "tests", "golden"), (3.0, 3.5, 4.0))
    argv = ["--mode", "test", "--root_dataset", root, "--model_dir", root, "--using_ckpts", "true", "--frames_per_step", "16"]
    rcv_main(argv)  # warm (kernels, allocator)
    pr = cProfile.Profile()
    pr.enable()
    res = rcv_main(argv)
    pr.disable()
    print("frame_loop_s", res[cls]["frame_loop_s"], "frames/s", n / res[cls]["frame_loop_s"], "net ms", res[cls]["net_time_ms"], "acc ms", res[cls]["acc_time_ms"])
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
