"""rcvpose_amd -- MI355X-native (gfx950) implementation of the RCVPose inference hot path.

The package holds the hand-written HIP kernels (csrc/, built into librcvpose_hip.so behind the C ABI
of include/rcvpose_hip.h) and the Python mirrors of the reference callables that `--mode=test`
traverses: models.fcnresnet.DenseFCNResNet152, AccumulatorSpace.{FCResBackbone, rgbd_to_point_cloud,
Accumulator_3D, estimate_6d_pose_lm}, utils.load_checkpoint, train.Trainer.Test.
"""
__version__ = "0.1.0"
