"""`main.py --mode=test` of the reference (reference main.py:13-74) for the MI355X build.

Same flags as the reference plus `--using_ckpts` (which the reference's test path reads, AccumulatorSpace.py:520,
but main.py forgot to define) and `--dtype`."""
import argparse

from .AccumulatorSpace import default_eval_threads
from .train import Trainer
from .utils import get_config, str2bool


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--mode', type=str, default='test', choices=['train', 'test'])
    p.add_argument('--gpu_id', type=int, default=-1)
    p.add_argument('--dname', type=str, default='lm', choices=['lm', 'ycb'])
    p.add_argument('--root_dataset', type=str, default='./datasets/LINEMOD')
    p.add_argument('--resume_train', type=str2bool, default=False)
    p.add_argument('--optim', type=str, default='Adam', choices=['Adam', 'SGD'])
    p.add_argument('--batch_size', type=str, default='4')
    p.add_argument('--class_name', type=str, default='ape')
    p.add_argument('--initial_lr', type=float, default=1e-4)
    p.add_argument('--kpt_num', type=str, default='1')
    p.add_argument('--model_dir', type=str, default='ckpts/')
    p.add_argument('--demo_mode', type=str2bool, default=False)
    p.add_argument('--test_occ', type=str2bool, default=False)
    p.add_argument('--using_ckpts', type=str2bool, default=True)
    p.add_argument('--dtype', type=str, default='f16x3', choices=['f16x3', 'fp32', 'bf16', 'fp16'],
                   help="engine arithmetic: f16x3 (default; fp16 hi/lo pairs, 3 MFMAs per product) and fp32 reproduce the reference's "
                        "fp32 forward to 1e-4; bf16 / fp16 are throughput modes outside that tolerance")
    p.add_argument('--frames_per_step', type=int, default=53,
                   help="frames of a class that travel through the device pipeline together (LINEMOD loop with checkpoints; default: "
                        "bench.py's batch); results do not depend on it")
    p.add_argument('--eval_threads', type=int, default=default_eval_threads(),
                   help="host threads (one HIP stream each) for the per-frame Horn / ADD(-S) / ICP evaluation of a batch")
    return p


def main(argv=None):
    opts = build_parser().parse_args(argv)
    opts.cfg = get_config()[1]
    if opts.mode != 'test':
        raise SystemExit("rcvpose_amd implements --mode=test (the inference hot path); use the reference for training")
    trainer = Trainer([], opts, [])
    return trainer.Test()


if __name__ == "__main__":
    main()
