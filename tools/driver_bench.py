#!/usr/bin/env python3
"""`main.py --mode=test` on a synthetic LINEMOD class with checkpoints: frames per second of the kept entry
point's frame loop (JPEG / .dpt decode on host threads, FCN -> seam -> voting in batches, Horn, ADD(-S), ICP per frame) next to
the bare FramePipeline on the same frames.

    python tools/driver_bench.py [--frames 64] [--frames_per_step 16] [--out gpurun_out/driver_bench.json]
"""
import argparse
import json
import os
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--frames_per_step", type=int, default=53)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "driver_bench.json"))
    ap.add_argument("--sweep", default="", help="comma-separated frames_per_step values: only the batched main.py runs, one per value")
    args = ap.parse_args()
    import test_gpu_driver as T  # the synthetic LINEMOD layout of the driver tests
    from rcvpose_amd import AccumulatorSpace as A
    from rcvpose_amd import synthetic
    from rcvpose_amd.main import main as rcv_main
    from rcvpose_amd.pipeline import FramePipeline

    with tempfile.TemporaryDirectory() as tmp:
        root = tmp + "/"
        cls, _ = T._write_dataset(root, n_frames=args.frames)
        T._write_lm_checkpoints(root, cls, os.path.join(ROOT, "tests", "golden"), (3.0, 3.5, 4.0))
        out = {"frames": args.frames, "frames_per_step": args.frames_per_step}
        if args.sweep:
            for n in [int(v) for v in args.sweep.split(",")]:
                res = rcv_main(["--mode", "test", "--root_dataset", root, "--model_dir", root, "--using_ckpts", "true", "--frames_per_step", str(n)])
                out["sweep_%d" % n] = {"frames_per_s": args.frames / res[cls]["frame_loop_s"], "net_time_ms": res[cls]["net_time_ms"],
                                       "acc_time_ms": res[cls]["acc_time_ms"]}
            json.dump(out, open(args.out, "w"), indent=1)
            print(json.dumps(out))
            return
        for n in (args.frames_per_step, 1):
            t0 = time.perf_counter()
            res = rcv_main(["--mode", "test", "--root_dataset", root, "--model_dir", root, "--using_ckpts", "true",
                            "--frames_per_step", str(n)])
            wall = time.perf_counter() - t0
            out["main_test_frames_per_step_%d" % n] = {"frame_loop_s": res[cls]["frame_loop_s"], "frames_per_s": args.frames / res[cls]["frame_loop_s"],
                                                       "wall_s_with_checkpoint_loading": wall, "net_time_ms": res[cls]["net_time_ms"],
                                                       "acc_time_ms": res[cls]["acc_time_ms"]}
        # the bare device pipeline on the same frames (inputs resident, no ADD / ICP)
        sds = []
        from rcvpose_amd import utils
        from rcvpose_amd.models.fcnresnet import DenseFCNResNet152

        for i in range(1, 4):
            m = DenseFCNResNet152(3, 2)
            m, _, _, _ = utils.load_checkpoint(m, None, root + cls + "_pt%d.pth.tar" % i)
            sds.append(m.state_dict())
        B = args.frames_per_step
        pipe = FramePipeline(sds, 480, 640, dtype="f16x3", frames_per_step=B)
        fr = [synthetic.synthetic_frame(i) for i in range(B)]
        rgb = torch.from_numpy(np.stack([f["rgb"] for f in fr])).cuda()
        depth = torch.from_numpy(np.stack([f["depth"] for f in fr]).view(np.int16)).cuda()
        rmax = np.array([4.6, 4.5, 4.5])
        pipe.enqueue_step(rgb, depth, rmax)
        pipe.finish()
        t0 = time.perf_counter()
        for _ in range(4):
            pipe.enqueue_step(rgb, depth, rmax)
        pipe.finish()
        out["frame_pipeline_frames_per_s"] = 4 * B / (time.perf_counter() - t0)
    out["main_over_pipeline"] = out["frame_pipeline_frames_per_s"] / out["main_test_frames_per_step_%d" % args.frames_per_step]["frames_per_s"]
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
