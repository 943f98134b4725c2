"""`models.fcnresnet` of the reference, re-implemented for MI355X.

Public surface kept from the reference (reference models/fcnresnet.py:6-45, :48-191):

    Bottleneck(in_channels, channels, stride=1, upsample=False)
    DenseFCNResNet152(input_channels=3, output_channels=2).forward(x) -> (seg_pred, radial_pred)
    ResFCNResNet152(input_channels=3, output_channels=2).forward(x)   -> (seg_pred, radial_pred)   (:194-318)

The modules hold parameters under exactly the reference's 1 307 state-dict keys (including the
`upsample_` branches the reference constructs in every Bottleneck but applies in only four, and the
BatchNorm `num_batches_tracked` buffers), so `load_state_dict(strict=True)` of a reference checkpoint
works, `module.`-prefixed (DataParallel) checkpoints included.  The forward itself does not run
torch operators: it hands the folded weights to the native engine (rcvpose_amd/fcn_engine.py ->
librcvpose_hip.so) whose hand-written MFMA kernels execute the whole graph.  Inference only; there
is no CPU fallback.
"""
import torch
import torch.nn as nn

from .. import fcn_engine


def _conv_bn(cin, cout, k, stride=1, padding=0, bias=False):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=padding, bias=bias), nn.BatchNorm2d(cout)


class Bottleneck(nn.Module):
    """1x1 -> 3x3 (carries the stride and a bias) -> 1x1 (x4), residual add, ReLU.  Parameter holder."""

    expansion = 4

    def __init__(self, in_channels, channels, stride=1, upsample=False):
        super().__init__()
        out_channels = channels * self.expansion
        self.conv1, self.bn1 = _conv_bn(in_channels, channels, 1)
        self.conv2, self.bn2 = _conv_bn(channels, channels, 3, stride=stride, padding=1, bias=True)
        self.conv3, self.bn3 = _conv_bn(channels, out_channels, 1)
        self.upsample = upsample
        # present in every block of the reference (and of its checkpoints), used only when upsample=True
        self.upsample_ = nn.Sequential(*_conv_bn(in_channels, out_channels, 1, stride=stride))
        self.relu = nn.ReLU(inplace=True)
        self.stride = stride

    def forward(self, x):
        raise NotImplementedError("Bottleneck is executed inside the native FCN engine; call DenseFCNResNet152.forward")


def _decoder_stage(cin, cout):
    conv, bn = _conv_bn(cin, cout, 3, padding=1, bias=True)
    return nn.Sequential(conv, bn, nn.ReLU(inplace=True))


class DenseFCNResNet152(nn.Module):
    # width, number of plain bottlenecks after the projecting one, stride of the projecting one
    _encoder = ((64, 2, 1), (128, 7, 2), (256, 35, 2), (512, 2, 2))
    _arch = "dense"

    def __init__(self, input_channels=3, output_channels=2):
        super().__init__()
        if input_channels != 3:
            # the reference accepts the argument but hard-wires conv1 to 3 inputs (fcnresnet.py:54)
            raise ValueError("the RCVPose FCN takes 3 input channels")
        self.input_channels = input_channels
        self.output_channels = output_channels
        self.conv1, self.bn1 = _conv_bn(3, 64, 7, stride=2, padding=3)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        cin = 64
        for idx, (width, plain, stride) in enumerate(self._encoder, start=1):
            setattr(self, "block%dup" % idx, Bottleneck(cin, width, stride=stride, upsample=True))
            cin = width * Bottleneck.expansion
            setattr(self, "block%d" % idx, nn.Sequential(*[Bottleneck(cin, width) for _ in range(plain)]))
        self.conv6, self.bn6 = _conv_bn(2048, 1024, 3, padding=1, bias=True)
        for name, cin_cat, cout in (("5", 2048 + 1024, 1024), ("4", 1024 + 1024, 512), ("3", 512 + 512, 256),
                                    ("2", 256 + 256, 128), ("1", 64 + 128, 64)):
            setattr(self, "conv_up" + name, _decoder_stage(cin_cat, cout))
            setattr(self, "up" + name, nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False))
        self.conv7 = _decoder_stage(64, 32)
        self.conv8 = nn.Conv2d(32, output_channels, kernel_size=1, stride=1)
        # "f16x3": fp16 hi / lo operand pairs, three MFMAs per product, fp32 accumulate -- matches the reference's fp32 forward to
        #          ~1e-5 relative like "fp32", several times faster (the default)
        # "fp32": exact-fp32 MFMA kernels (v_mfma_f32_16x16x4_f32)
        # "bf16" / "fp16": 16-bit storage / fp32 accumulate (throughput modes outside the 1e-4 tolerance)
        self.compute_dtype = "f16x3"
        self._engine = None
        self._engine_key = None

    def set_compute_dtype(self, name):
        if name not in ("f16x3", "fp32", "bf16", "fp16"):
            raise ValueError("compute dtype must be 'f16x3', 'fp32', 'bf16' or 'fp16'")
        self.compute_dtype = name
        return self

    def _stamp(self):
        v = 0
        for t in list(self.parameters()) + list(self.buffers()):
            v += t._version
        return v

    def engine(self, H, W):
        key = (H, W, self.compute_dtype, self._stamp())
        if self._engine is None or self._engine_key != key:
            self._engine = fcn_engine.FcnEngine([self.state_dict()], H, W, dtype=self.compute_dtype, arch=self._arch)
            # the module is the per-call (synchronous) boundary: half-precision storage modes fail loudly on overflow
            self._engine.check_finite = self.compute_dtype != "fp32"
            self._engine_key = key
        return self._engine

    def forward(self, x):
        """x: float32 [B,3,H,W] -> (seg_pred [B,1,H,W], radial_pred [B,1,H,W])  (reference :121-191)."""
        if self.training:
            raise NotImplementedError("rcvpose_amd implements the inference path (--mode=test); call model.eval()")
        if self.output_channels != 2:
            raise NotImplementedError("the native head emits the (seg, radial) pair, output_channels must be 2")
        if not x.is_cuda:
            raise RuntimeError("rcvpose_amd runs on MI355X only: move the input to 'cuda' (no CPU fallback)")
        with torch.no_grad():
            eng = self.engine(int(x.shape[2]), int(x.shape[3]))
            seg, radial = eng.forward(x.float())
        return seg[0], radial[0]


class ResFCNResNet152(DenseFCNResNet152):
    """The additive-skip variant of the reference (models/fcnresnet.py:194-318): same encoder, decoder stages
    `conv_upK(upK(x) + conv_upK_1(skip))`, a second `up1` in front of conv7 (128 -> 64) and conv8 (64 -> 2).  The reference
    imports it next to DenseFCNResNet152 but none of its drivers instantiates it.  Same 1 301 state-dict keys."""

    _arch = "res"

    def __init__(self, input_channels=3, output_channels=2):
        nn.Module.__init__(self)
        if input_channels != 3:
            raise ValueError("the RCVPose FCN takes 3 input channels")
        self.input_channels = input_channels
        self.output_channels = output_channels
        self.conv1, self.bn1 = _conv_bn(3, 64, 7, stride=2, padding=3)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        cin = 64
        for idx, (width, plain, stride) in enumerate(self._encoder, start=1):
            setattr(self, "block%dup" % idx, Bottleneck(cin, width, stride=stride, upsample=True))
            cin = width * Bottleneck.expansion
            setattr(self, "block%d" % idx, nn.Sequential(*[Bottleneck(cin, width) for _ in range(plain)]))
        for name, cskip, cup, cout in (("4", 1024, 2048, 1024), ("3", 512, 1024, 512), ("2", 256, 512, 256), ("1", 64, 256, 128)):
            setattr(self, "conv_up%s_1" % name, nn.Conv2d(cskip, cup, kernel_size=1))
            setattr(self, "conv_up" + name, _decoder_stage(cup, cout))
            setattr(self, "up" + name, nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False))
        self.conv7 = _decoder_stage(128, 64)
        self.conv8 = nn.Conv2d(64, output_channels, kernel_size=1, stride=1)
        self.compute_dtype = "f16x3"
        self._engine = None
        self._engine_key = None
