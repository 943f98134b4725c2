import cProfile, pstats, os, sys, tempfile, types
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
import test_gpu_driver as T
from rcvpose_amd import AccumulatorSpace as A
with tempfile.TemporaryDirectory() as tmp:
    root = tmp + "/"
    cls, _ = T._write_dataset(root, n_frames=16)
    T._write_lm_checkpoints(root, cls, os.path.join(ROOT, "tests", "golden"), (3.0, 3.5, 4.0))
    opts = types.SimpleNamespace(root_dataset=root, model_dir=root, using_ckpts=True, demo_mode=False, test_occ=False, class_names=[cls], frames_per_step=8)
    A.estimate_6d_pose_lm(opts)
    pr = cProfile.Profile(); pr.enable()
    A.estimate_6d_pose_lm(opts)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
