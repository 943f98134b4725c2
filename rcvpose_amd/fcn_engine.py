"""Layer program + weight packing for the native FCN executor (rcv_fcn_* in include/rcvpose_hip.h).

The topology restated here is the reference's DenseFCNResNet152 (reference models/fcnresnet.py:48-191,
Bottleneck :6-45).  Building the program and packing the weights is host-side plumbing (pure
Python / torch-CPU, testable without a GPU); every FLOP of the forward runs in the HIP kernels of
rcvpose_amd/csrc/fcn.hip.  There is no CPU fallback for the forward itself.
"""
import ctypes

import numpy as np

from . import _hip

# (first bottleneck, nn.Sequential of the rest, how many in the rest, width, stride of the first)
# reference models/fcnresnet.py:59-83
STAGES = (
    ("block1up", "block1", 2, 64, 1),
    ("block2up", "block2", 7, 128, 2),
    ("block3up", "block3", 35, 256, 2),
    ("block4up", "block4", 2, 512, 2),
)
BN_EPS = 1e-5
FUSE_SPAN = 3  # the executor fuses at most this many consecutive ops into one kernel (slot liveness, assign_slots)


class _Builder:
    def __init__(self):
        self.ops = []       # dicts
        self.tensors = []   # per logical tensor: elems per (group, image)
        self.wspecs = []    # per op: None or dict(conv=..., bn=..., cout=..., cin=..., k=...)
        self.named = {}     # stage name -> (tensor id, (h, w, c)); names follow the reference's module names

    def tensor(self, h, w, c):
        self.tensors.append(h * w * c)
        self.shapes = getattr(self, "shapes", [])
        self.shapes.append((h, w, c))
        return len(self.tensors) - 1

    def name(self, label, t):
        self.named[label] = (t, self.shapes[t])

    def add(self, kind, wspec=None, **kw):
        op = dict(kind=kind, src0=-1, src1=-1, res=-1, dst=-1, c0=0, c1=0, cout=0, hin=0, win=0, hout=0, wout=0,
                  ksize=1, stride=1, pad=0, relu=0)
        op.update(kw)
        self.ops.append(op)
        self.wspecs.append(wspec)
        return op


def _conv_op(b, src0, c0, cout, hin, win, k, stride, relu, wspec, src1=-1, c1=0, res=-1):
    pad = 1 if k == 3 else 0
    hout, wout = (hin + 2 * pad - k) // stride + 1, (win + 2 * pad - k) // stride + 1
    dst = b.tensor(hout, wout, cout)
    b.add(_hip.OP_CONV, wspec, src0=src0, src1=src1, res=res, dst=dst, c0=c0, c1=c1, cout=cout, hin=hin, win=win,
          hout=hout, wout=wout, ksize=k, stride=stride, pad=pad, relu=relu)
    return dst, hout, wout


def _encoder(b, H, W):
    """Stem, max-pool and the four ResNet-152 stages shared by both FCN variants (reference models/fcnresnet.py:54-83,
    :200-228).  Returns (stem tensor at H/2, [(tensor, channels, h, w) of block1..block4])."""
    h2, w2 = H // 2, W // 2
    x = b.tensor(h2, w2, 64)
    b.add(_hip.OP_STEM, dict(conv="conv1", bn="bn1", bias=False), src0=_hip.SLOT_INPUT, dst=x, c0=3, cout=64, hin=H,
          win=W, hout=h2, wout=w2, ksize=7, stride=2, pad=3, relu=1)
    b.name("bn1", x)
    h4, w4 = H // 4, W // 4
    cur = b.tensor(h4, w4, 64)
    b.add(_hip.OP_MAXPOOL, src0=x, dst=cur, c0=64, cout=64, hin=h2, win=w2, hout=h4, wout=w4, ksize=3, stride=2, pad=1)
    b.name("maxpool", cur)
    cin, hh, ww = 64, h4, w4
    skips = []

    def bottleneck(prefix, src, cin, width, hin, win, stride, project):
        t1, _, _ = _conv_op(b, src, cin, width, hin, win, 1, 1, 1, dict(conv=prefix + ".conv1", bn=prefix + ".bn1", bias=False))
        t2, ho, wo = _conv_op(b, t1, width, width, hin, win, 3, stride, 1,
                              dict(conv=prefix + ".conv2", bn=prefix + ".bn2", bias=True))
        res = src
        if project:  # `upsample_` is applied only where upsample=True (reference :35-36)
            res, _, _ = _conv_op(b, src, cin, 4 * width, hin, win, 1, stride, 0,
                                 dict(conv=prefix + ".upsample_.0", bn=prefix + ".upsample_.1", bias=False))
        out, _, _ = _conv_op(b, t2, width, 4 * width, ho, wo, 1, 1, 1,
                             dict(conv=prefix + ".conv3", bn=prefix + ".bn3", bias=False), res=res)
        return out, ho, wo

    for first, rest, count, width, stride in STAGES:
        cur, hh, ww = bottleneck(first, cur, cin, width, hh, ww, stride, True)
        cin = 4 * width
        for i in range(count):
            cur, hh, ww = bottleneck("%s.%d" % (rest, i), cur, cin, width, hh, ww, 1, False)
        skips.append((cur, cin, hh, ww))
        b.name(rest, cur)
    return x, skips


def build_program_res(H, W):
    """The op list of one ResFCNResNet152 forward (reference models/fcnresnet.py:263-318): additive skips
    `conv_upK(up(x) + conv_upK_1(skip))` -- the 1x1 lateral convolution takes the upsampled tensor as its residual --
    and a last x2 upsampling in front of conv7 (128 -> 64) and the head (64 -> 2)."""
    if H % 32 or W % 32 or H <= 0 or W <= 0:
        raise ValueError("ResFCNResNet152 needs H and W to be positive multiples of 32")
    b = _Builder()
    x, skips = _encoder(b, H, W)
    (x2s, c2, _, _), (x4s, c4, _, _), (x8s, c8, _, _), (x16s, c16, hh, ww) = skips
    up, cup = x16s, c16
    for name, (skip, cskip), cout in (("4", (x8s, c8), 1024), ("3", (x4s, c4), 512), ("2", (x2s, c2), 256), ("1", (x, 64), 128)):
        dst = b.tensor(hh * 2, ww * 2, cup)
        b.add(_hip.OP_UPSAMPLE2, src0=up, dst=dst, c0=cup, cout=cup, hin=hh, win=ww, hout=hh * 2, wout=ww * 2)
        hh, ww = hh * 2, ww * 2
        lat, _, _ = _conv_op(b, skip, cskip, cup, hh, ww, 1, 1, 0, dict(conv="conv_up%s_1" % name, bn=None, bias=True), res=dst)
        up, _, _ = _conv_op(b, lat, cup, cout, hh, ww, 3, 1, 1, dict(conv="conv_up%s.0" % name, bn="conv_up%s.1" % name, bias=True))
        b.name("conv_up" + name, up)
        cup = cout
    dst = b.tensor(hh * 2, ww * 2, cup)
    b.add(_hip.OP_UPSAMPLE2, src0=up, dst=dst, c0=cup, cout=cup, hin=hh, win=ww, hout=hh * 2, wout=ww * 2)
    hh, ww = hh * 2, ww * 2
    up, _, _ = _conv_op(b, dst, cup, 64, hh, ww, 3, 1, 1, dict(conv="conv7.0", bn="conv7.1", bias=True))
    b.name("conv7", up)
    b.add(_hip.OP_HEAD, dict(conv="conv8", bn=None, bias=True), src0=up, dst=_hip.SLOT_OUTPUT, c0=64, cout=2, hin=hh,
          win=ww, hout=hh, wout=ww)
    return b


def build_program(H, W):
    """The op list of one DenseFCNResNet152 forward on an HxW image (H, W multiples of 32)."""
    if H % 32 or W % 32 or H <= 0 or W <= 0:
        raise ValueError("DenseFCNResNet152 needs H and W to be positive multiples of 32 (skip concatenations)")
    b = _Builder()
    x, skips = _encoder(b, H, W)

    def conv(src0, c0, cout, hin, win, k, stride, relu, wspec, src1=-1, c1=0, res=-1):
        return _conv_op(b, src0, c0, cout, hin, win, k, stride, relu, wspec, src1=src1, c1=c1, res=res)

    (x2s, c2, _, _), (x4s, c4, _, _), (x8s, c8, _, _), (x16s, c16, h16, w16) = skips
    up, _, _ = conv(x16s, c16, 1024, h16, w16, 3, 1, 1, dict(conv="conv6", bn="bn6", bias=True))
    b.name("bn6", up)
    cup, hh, ww = 1024, h16, w16
    for name, (skip, cskip), cout in (("5", (x16s, c16), 1024), ("4", (x8s, c8), 512), ("3", (x4s, c4), 256),
                                      ("2", (x2s, c2), 128), ("1", (x, 64), 64)):
        up, _, _ = conv(up, cup, cout, hh, ww, 3, 1, 1, dict(conv="conv_up%s.0" % name, bn="conv_up%s.1" % name, bias=True),
                        src1=skip, c1=cskip)
        b.name("conv_up" + name, up)
        dst = b.tensor(hh * 2, ww * 2, cout)
        b.add(_hip.OP_UPSAMPLE2, src0=up, dst=dst, c0=cout, cout=cout, hin=hh, win=ww, hout=hh * 2, wout=ww * 2)
        up, cup, hh, ww = dst, cout, hh * 2, ww * 2
        b.name("up" + name, up)
    up, _, _ = conv(up, 64, 32, hh, ww, 3, 1, 1, dict(conv="conv7.0", bn="conv7.1", bias=True))
    b.name("conv7", up)
    b.add(_hip.OP_HEAD, dict(conv="conv8", bn=None, bias=True), src0=up, dst=_hip.SLOT_OUTPUT, c0=32, cout=2, hin=hh,
          win=ww, hout=hh, wout=ww)
    return b


def assign_slots(builder, reuse=True):
    """Map logical tensors to reusable workspace slots (greedy, by last use) so that the working
    set of consecutive layers stays small enough to live in the 256 MB Infinity Cache.
    reuse=False keeps one slot per tensor (stage-by-stage debugging)."""
    if not reuse:
        return [dict(op) for op in builder.ops], list(builder.tensors)
    last_use = {}
    for i, op in enumerate(builder.ops):
        for key in ("src0", "src1", "res", "dst"):
            t = op[key]
            if t >= 0:
                last_use[t] = i
    slot_of, slot_size, free = {}, [], []
    for i, op in enumerate(builder.ops):
        t = op["dst"]
        if t >= 0 and t not in slot_of:
            size = builder.tensors[t]
            cand = [s for s in free if slot_size[s] == size]
            if cand:
                s = cand[0]
                free.remove(s)
            else:
                s = len(slot_size)
                slot_size.append(size)
            slot_of[t] = s
        # A slot is released two ops after the last reader: the executor may run up to three consecutive ops as one kernel
        # (3x3 -> expand -> next reduce, upsample -> conv7 -> head), in which later tiles still read an input while earlier
        # tiles already write the outputs of the following ops.
        for j in range(max(0, i - FUSE_SPAN + 1), i - FUSE_SPAN + 2):
            if j < 0:
                continue
            for key in ("src0", "src1", "res"):
                t = builder.ops[j][key]
                if t >= 0 and last_use[t] == j and slot_of[t] not in free:
                    free.append(slot_of[t])
    ops = []
    for op in builder.ops:
        o = dict(op)
        for key in ("src0", "src1", "res", "dst"):
            if o[key] >= 0:
                o[key] = slot_of[o[key]]
        ops.append(o)
    return ops, slot_size


def fold_conv_bn(sd, spec, cout):
    """Fold eval-mode BatchNorm into the preceding convolution (float64 on the host).

    y = gamma * (conv(x) + b - mean) / sqrt(var + eps) + beta  ->  conv'(x) + b'
    Returns (weight [cout, kh, kw, cin] float64, bias [cout] float64)."""
    import torch

    w = sd[spec["conv"] + ".weight"].detach().to("cpu", torch.float64)
    b = sd[spec["conv"] + ".bias"].detach().to("cpu", torch.float64) if spec["bias"] else torch.zeros(cout, dtype=torch.float64)
    if spec["bn"] is not None:
        p = spec["bn"]
        gamma = sd[p + ".weight"].detach().to("cpu", torch.float64)
        beta = sd[p + ".bias"].detach().to("cpu", torch.float64)
        mean = sd[p + ".running_mean"].detach().to("cpu", torch.float64)
        var = sd[p + ".running_var"].detach().to("cpu", torch.float64)
        scale = gamma / torch.sqrt(var + BN_EPS)
        w = w * scale.view(-1, 1, 1, 1)
        b = (b - mean) * scale + beta
    return w.permute(0, 2, 3, 1).contiguous(), b


def split_halves(w):
    """float64 tensor -> (hi, lo) float16 with hi = half(w), lo = half(w - hi): hi + lo carries 22 significand bits."""
    import torch

    hi = w.to(torch.float16)
    lo = (w - hi.to(torch.float64)).to(torch.float16)
    return hi, lo


def pack_split(w, kind):
    """One network's BN-folded float64 weights [cout, kh, kw, cin] in the RCV_F16X3 layout (include/rcvpose_hip.h):
    every output channel is scaled by a power of two 2^k that puts its largest magnitude into [2^13, 2^14) -- the lo halves of
    all weights that matter are then normal half-precision numbers, measured to be worth 20x in accuracy on small weights
    (tools/probe/split_probe.hip) -- and the epilogue multiplies by 2^-k, exactly.
    -> (flat float16 weights, float32 scales [cout])."""
    import torch

    cout = w.shape[0]
    mx = w.reshape(cout, -1).abs().amax(dim=1)
    _, e = torch.frexp(mx)  # mx = m * 2^e with m in [0.5, 1)
    k = torch.where(mx > 0, 14 - e, torch.zeros_like(e)).to(torch.float64)
    ws = w * torch.exp2(k).view(-1, 1, 1, 1)
    if not bool(torch.isfinite(mx).all()):
        raise ValueError("non-finite weights after BatchNorm folding")
    hi, lo = split_halves(ws)  # (finite by construction: the scaled magnitudes are below 2^14)
    if kind == _hip.OP_CONV:
        cin = w.shape[3]
        if cin % 32:
            raise ValueError("RCV_F16X3 convolutions need a multiple of 32 input channels")
        # [cout][kh][kw][cin/32][hi 32 | lo 32]
        flat = torch.stack([hi.reshape(cout, w.shape[1], w.shape[2], cin // 32, 32),
                            lo.reshape(cout, w.shape[1], w.shape[2], cin // 32, 32)], dim=4).reshape(-1)
    else:  # stem / head: all hi halves, then all lo halves
        flat = torch.cat([hi.reshape(-1), lo.reshape(-1)])
    return flat, torch.exp2(-k).to(torch.float32)


def q6_encode(v):
    """Blocks of 16 values (last axis; any float tensor whose values are exact in float64) -> (codes uint8 [..., 16], scale uint8 [...]):
    the e2m3 block format of v_mfma_scale_f32_16x16x128_f8f6f4 as the engine uses it (include/rcvpose_hip.h, option 11): scale
    2^(floor(log2 max|block|) - 2) as an e8m0 byte (an all-zero block: byte 1), codes sign | 2-bit exponent (bias 1) | 3-bit mantissa,
    round to nearest even, saturating at 7.5 -- what v_cvt_scalef32_pk32_fp6_f16 produces (tools/probe/fp6_mfma_probe.hip)."""
    import torch

    v = v.to(torch.float64)
    amax = v.abs().amax(dim=-1)
    _, e = torch.frexp(amax)                              # amax = m 2^e, m in [0.5, 1)
    biased = torch.where(amax > 0, e - 1 + 127, torch.zeros_like(e))
    ex = torch.clamp(biased, min=3) - 2
    y = v / torch.exp2((ex - 127).to(torch.float64)).unsqueeze(-1)
    a = y.abs()
    _, ea = torch.frexp(a)                                # binade of a: [2^(ea-1), 2^ea)
    step = torch.exp2(torch.clamp(ea - 1, min=0).to(torch.float64) - 3)  # 1/8 below 2, 1/4 in [2,4), 1/2 in [4,8)
    q = torch.clamp(torch.round(a / step) * step, max=7.5)
    _, eq = torch.frexp(q)
    ebits = torch.where(q >= 1, eq, torch.zeros_like(eq))          # 1 .. 3 for [1,2), [2,4), [4,8); 0 below 1
    mant = torch.where(q >= 1, (q / torch.exp2((ebits - 1).to(torch.float64)) - 1) * 8, q * 8)
    sign = (torch.signbit(v)).to(torch.int64)
    codes = (sign << 5) | (ebits.to(torch.int64) << 3) | mant.round().to(torch.int64)
    return codes.to(torch.uint8), ex.to(torch.uint8)


def q6_decode(codes, scale):
    """Inverse of q6_encode -> float64 values (tests, emulation)."""
    import torch

    c = codes.to(torch.int64)
    e, m = (c >> 3) & 3, (c & 7).to(torch.float64)
    mag = torch.where(e == 0, m / 8, (1 + m / 8) * torch.exp2((e - 1).to(torch.float64)))
    val = torch.where(((c >> 5) & 1) == 1, -mag, mag)
    return val * torch.exp2(scale.to(torch.float64) - 127).unsqueeze(-1)


def q6_chunk(codes, scale, scale_byte):
    """16 codes + the block scale -> the 16-byte chunk of a q6 row: element j in bits [6j, 6j + 6), scale in byte `scale_byte`
    (12: activations, 14: weights), the other bytes of the last dword zero."""
    import torch

    c = codes.to(torch.int64).reshape(*codes.shape[:-1], 4, 4)
    word = c[..., 0] | (c[..., 1] << 6) | (c[..., 2] << 12) | (c[..., 3] << 18)   # 24 bits per four codes
    b = torch.stack([word & 255, (word >> 8) & 255, (word >> 16) & 255], dim=-1).reshape(*codes.shape[:-1], 12)
    tail = torch.zeros(*codes.shape[:-1], 4, dtype=torch.int64)
    tail[..., scale_byte - 12] = scale.to(torch.int64)
    return torch.cat([b, tail], dim=-1).to(torch.uint8)


def pack_split_q6(w):
    """One network's BN-folded weights [cout, kh, kw, cin] of a 3x3 convolution as q6 rows (engine option 11): per (cout, tap, 32
    channels) [32 hi halves | lo6[0:16] | lo6[16:32] | hi6[0:16] | hi6[16:32]], hi / lo the halves of pack_split (same per-channel
    power-of-two scale, so the op's [bias | scale] record serves both copies).  -> flat float16 (bit patterns)."""
    import torch

    cout, kh, kw, cin = w.shape
    mx = w.reshape(cout, -1).abs().amax(dim=1)
    _, e = torch.frexp(mx)
    k = torch.where(mx > 0, 14 - e, torch.zeros_like(e)).to(torch.float64)
    hi, lo = split_halves(w * torch.exp2(k).view(-1, 1, 1, 1))
    hi = hi.reshape(cout, kh, kw, cin // 32, 32)
    lo = lo.reshape(cout, kh, kw, cin // 32, 32)
    chunks = []
    for src in (lo, lo, hi, hi):
        half = len(chunks) & 1
        blk = src[..., 16 * half:16 * half + 16]
        codes, sc = q6_encode(blk)
        chunks.append(q6_chunk(codes, sc, 14))
    rows = torch.cat([hi.contiguous().view(torch.uint8).reshape(cout, kh, kw, cin // 32, 64)] + chunks, dim=-1)  # [..., 128] bytes
    return rows.reshape(-1).view(torch.float16)


Q6_DEFAULT = r"block[234]\.\d+\.conv2"


def q6_wanted(spec, op, pattern=Q6_DEFAULT):
    """The ops that get a q6 copy of their weights (fp6 cross terms, engine option 11): the 3x3 / stride-1 convolutions of the
    bottleneck blocks of layers 2-4 (blockN.M.conv2 with >= 128 channels: 44 launches, 20 ms of a 110 ms forward at the bench
    batch).  Their rounding noise enters the residual trunk through the following 1x1 and is diluted there: measured map error with
    all of them in q6 1.6-3.6e-5 at 480x640 (three-MFMA engine: 0.4-0.9e-5).  A decoder convolution passes its error straight to
    the maps (1-4e-5 EACH; all seven: 6e-5 and more) -- they keep three MFMAs (tools/probe/fp8_cross_probe.py --each / --enc)."""
    import re

    return (op["kind"] == _hip.OP_CONV and op["ksize"] == 3 and op["stride"] == 1 and op["src1"] < 0 and op["c0"] % 32 == 0 and
            op["cout"] % 128 == 0 and re.fullmatch(pattern, spec["conv"]) is not None)


# bilinear x2 (align_corners=False) seen from the 3x3 convolution that follows it: the high-resolution tap t - 1 (t = 0, 1, 2) of output
# row 2i + p reads low-resolution rows i - 1, i, i + 1 with these weights (row p, then t, then k).  out[2i] = .25 in[i-1] + .75 in[i],
# out[2i+1] = .75 in[i] + .25 in[i+1] (reference models/fcnresnet.py:93,177: nn.Upsample(scale_factor=2, mode='bilinear'))
UP2_TAPS = (((0.75, 0.25, 0.0), (0.25, 0.75, 0.0), (0.0, 0.75, 0.25)),
            ((0.25, 0.75, 0.0), (0.0, 0.75, 0.25), (0.0, 0.25, 0.75)))


def compose_upsample_conv3(w):
    """BN-folded float64 weights [cout, 3, 3, cin] of a 3x3 / pad-1 convolution that reads a x2 bilinear upsampling -> the weights
    [4 * cout, 3, 3, cin] of the equivalent 3x3 convolution on the LOW-resolution tensor whose output channel (2 py + px) * cout + co
    is channel co of the high-resolution pixel (2 i + py, 2 j + px) -- both maps are linear, the composition is exact.  (Away from
    the image border: the upsampling clamps there and the convolution pads with zeros, so the outermost ring of high-resolution
    pixels is recomputed separately: k_tail_x3 on the border tiles.)"""
    import torch

    cout = w.shape[0]
    C = torch.tensor(UP2_TAPS, dtype=torch.float64)                      # [parity, t, k]
    # Wc[py, px, co, a, b, ci] = sum_{r, s} w[co, r, s, ci] C[py, r, a] C[px, s, b]
    wc = torch.einsum("orsc,pra,qsb->pqoabc", w.to(torch.float64), C, C)
    return wc.reshape(4 * cout, 3, 3, w.shape[3]).contiguous()


def tail_composed_wanted(builder, i):
    """Op i is the conv7 of a (x2 upsampling -> 3x3 convolution to 32 channels -> head) tail: the executor can run the three as ONE
    3x3 convolution on the low-resolution tensor with 4 x 32 parity-stacked output channels + the head in its epilogue (engine
    option 13), given the composed weights as the op's second copy."""
    ops = builder.ops
    if i < 1 or i + 1 >= len(ops):
        return False
    u, c, d = ops[i - 1], ops[i], ops[i + 1]
    return (u["kind"] == _hip.OP_UPSAMPLE2 and c["kind"] == _hip.OP_CONV and d["kind"] == _hip.OP_HEAD and c["src0"] == u["dst"] and
            c["src1"] < 0 and c["res"] < 0 and c["ksize"] == 3 and c["stride"] == 1 and c["pad"] == 1 and c["cout"] == 32 and
            c["c0"] % 32 == 0 and d["src0"] == c["dst"] and d["c0"] == 32 and d["cout"] == 2 and c["relu"] == 1)


def pack_weights(state_dicts, builder, ops, dtype, q6=False):
    """-> (weights tensor [engine dtype], bias tensor fp32, ops with w_off / b_off / wq_off filled)."""
    import torch

    sds = [{(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()} for sd in state_dicts]
    tdt = {_hip.RCV_F32: torch.float32, _hip.RCV_BF16: torch.bfloat16, _hip.RCV_F16: torch.float16,
           _hip.RCV_F16X3: torch.float16}[dtype]
    wparts, bparts, woff, boff = [], [], 0, 0
    for idx, (op, spec) in enumerate(zip(ops, builder.wspecs)):
        op["wq_off"] = 0
        if spec is None:
            op["w_off"], op["b_off"] = 0, 0
            continue
        ws, bs, wq, bq = [], [], [], []
        want_q6 = bool(q6) and dtype == _hip.RCV_F16X3 and q6_wanted(spec, op, Q6_DEFAULT if q6 is True else q6)
        want_tailc = dtype == _hip.RCV_F16X3 and tail_composed_wanted(builder, idx)
        for sd in sds:
            w, b = fold_conv_bn(sd, spec, op["cout"])
            wmax = float(w.abs().max()) if w.numel() else 0.0
            if not np.isfinite(wmax) or (dtype == _hip.RCV_F16 and wmax > 65504.0):
                raise ValueError("weights of %s are not finite in the engine's storage type (|w| max %g)" % (spec["conv"], wmax))
            if dtype == _hip.RCV_F16X3:  # per network: [weights], [bias | scale]
                flat, scale = pack_split(w, op["kind"])
                ws.append(flat)
                bs.append(torch.cat([b.to(torch.float32), scale]))
                if want_q6:
                    wq.append(pack_split_q6(w))
                if want_tailc:  # the parity-composed weights [4 * cout][3][3][cin] and their [bias x 4 | scale] record
                    flat_c, scale_c = pack_split(compose_upsample_conv3(w), op["kind"])
                    wq.append(flat_c)
                    bq.append(torch.cat([b.to(torch.float32).repeat(4), scale_c]))
            else:
                ws.append(w.reshape(-1))
                bs.append(b)
        wcat = torch.cat(ws).to(tdt)
        bcat = torch.cat(bs + bq).to(torch.float32)  # (second-copy records of all networks BEHIND the first-copy records: strides unchanged)
        op["w_off"], op["b_off"] = woff, boff
        pad = (-wcat.numel()) % 64  # keep every op's weights 128-byte aligned for 16-byte vector loads
        if pad:
            wcat = torch.cat([wcat, torch.zeros(pad, dtype=tdt)])
        padb = (-bcat.numel()) % 4
        if padb:
            bcat = torch.cat([bcat, torch.zeros(padb, dtype=torch.float32)])
        wparts.append(wcat)
        bparts.append(bcat)
        woff += wcat.numel()
        boff += bcat.numel()
        if wq:  # the second copy of all groups (q6 rows: same strides as the first copy; parity-composed tail: four times the rows)
            qcat = torch.cat(wq)
            assert qcat.numel() % 64 == 0 and (want_tailc or qcat.numel() + pad == wcat.numel())
            op["wq_off"] = woff
            wparts.append(qcat)
            woff += qcat.numel()
    return torch.cat(wparts), torch.cat(bparts), ops


def conv_flops(ops):
    """Algorithmic FLOPs of the convolutions of one forward (per group, per image): 2*M*N*K."""
    fl = 0
    for o in ops:
        if o["kind"] in (_hip.OP_CONV, _hip.OP_STEM, _hip.OP_HEAD):
            fl += 2 * o["hout"] * o["wout"] * o["cout"] * o["ksize"] ** 2 * (o["c0"] + (o["c1"] if o["src1"] >= 0 else 0))
    return fl


class FcnEngine:
    """`groups` DenseFCNResNet152 networks (one state dict each) sharing one input image.

    forward(x): x float32 [B,3,H,W] on the GPU -> (seg, radial), each float32 [G,B,1,H,W].
    dtype: storage type of weights and activations (fp32 accumulation throughout): "fp32" (parity mode on the fp32 matrix
    instruction), "f16x3" (the FAST parity mode: half-precision hi / lo pairs, three MFMAs per product -- maps within 1e-4 of
    the reference like fp32), "bf16" (the throughput mode BASELINE.json names) or "fp16" (IEEE half: the same kernels and
    speed as bf16 with 11 instead of 8 significand bits).
    q6 (f16x3 only; default off): True or a regex over convolution names -- those stride-1 3x3 convolutions run the cross terms of
    the split product as one block-scaled fp6 matrix instruction (engine option 11, q6_wanted): -18 % on those launches, -3 % on
    the forward, for 2-3x the map error (worst network of the bench 3.4e-5 -> 7.8e-5 of the 1e-4 tolerance: r05 measurement).
    """

    def __init__(self, state_dicts, H, W, dtype="bf16", device="cuda", reuse_slots=True, arch="dense", q6=False):
        import torch

        self.lib = _hip.lib()
        self.H, self.W = int(H), int(W)
        self.groups = len(state_dicts)
        self.dtype = {"bf16": _hip.RCV_BF16, "fp32": _hip.RCV_F32, "f32": _hip.RCV_F32, "fp16": _hip.RCV_F16, "f16": _hip.RCV_F16,
                      "f16x3": _hip.RCV_F16X3, "fp16x3": _hip.RCV_F16X3}[dtype]
        self.dtype_name = {_hip.RCV_BF16: "bf16", _hip.RCV_F32: "fp32", _hip.RCV_F16: "fp16", _hip.RCV_F16X3: "f16x3"}[self.dtype]
        if arch not in ("dense", "res"):
            raise ValueError("arch must be 'dense' (DenseFCNResNet152) or 'res' (ResFCNResNet152)")
        self.arch = arch
        builder = build_program(self.H, self.W) if arch == "dense" else build_program_res(self.H, self.W)
        ops, slot_sizes = assign_slots(builder, reuse=reuse_slots)
        self.stages = builder.named if not reuse_slots else {}
        w, b, ops = pack_weights(state_dicts, builder, ops, self.dtype, q6=q6)
        self.ops = ops
        self.slot_sizes = slot_sizes
        self.weights = w.to(device)
        self.bias = b.to(device)
        arr = (_hip.Op * len(ops))()
        for i, o in enumerate(ops):
            for f, _ in _hip.Op._fields_:
                setattr(arr[i], f, int(o[f]))
        slots = (ctypes.c_int64 * len(slot_sizes))(*slot_sizes)
        handle = ctypes.c_void_p()
        _hip.check(self.lib.rcv_fcn_create(arr, len(ops), len(slot_sizes), slots, self.groups, self.dtype,
                                           _hip.ptr(self.weights), _hip.ptr(self.bias), ctypes.byref(handle)),
                   "rcv_fcn_create")
        self.handle = handle
        self.device = device
        self._ws = None
        self._ws_batch = 0
        self.radial_f16 = False
        # half-precision storage has a range: f16x3 clamps magnitudes beyond 65504 (finite and wrong), fp16 turns them into
        # infinities.  The kernels record both in the engine's sticky flag word (RCV_E_RANGE, include/rcvpose_hip.h).  With
        # check_finite the forward synchronises, reads that word (range_status) and the maps' finiteness, and raises; the
        # module boundary (models.fcnresnet, FCResBackbone) switches it on.  Batched pipelines stay asynchronous: the next
        # forward after a flagged one fails with RCV_E_RANGE, and FramePipeline.read_results() checks the word itself.
        self.check_finite = False
        self.flops_per_image_group = conv_flops(ops)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.rcv_fcn_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def workspace_bytes(self, B):
        """Device bytes of the activation workspace for a batch of B images (nothing is allocated)."""
        return int(self.lib.rcv_fcn_workspace_bytes(self.handle, B))

    def workspace(self, B):
        import torch

        if self._ws is None or self._ws_batch < B:
            nbytes = int(self.lib.rcv_fcn_workspace_bytes(self.handle, B))
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws_batch = B
        return self._ws

    def forward(self, x, out=None):
        import torch

        if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("x must be a CUDA float32 [B,3,H,W] tensor")
        if tuple(x.shape[2:]) != (self.H, self.W):
            raise ValueError("engine was built for %dx%d input" % (self.H, self.W))
        x = x.contiguous()
        B = x.shape[0]
        ws = self.workspace(B)
        if out is None:
            seg = torch.empty((self.groups, B, 1, self.H, self.W), dtype=torch.float32, device=x.device)
            radial = torch.empty_like(seg, dtype=torch.float16 if self.radial_f16 else torch.float32)
        else:
            seg, radial = out
            if radial.dtype != (torch.float16 if self.radial_f16 else torch.float32) or seg.dtype != torch.float32:
                raise ValueError("output planes: seg float32, radial %s" % ("float16 (option 8)" if self.radial_f16 else "float32"))
        _hip.check(self.lib.rcv_fcn_forward(self.handle, _hip.ptr(x), B, _hip.ptr(seg), _hip.ptr(radial), _hip.ptr(ws),
                                            ws.numel(), _hip.current_stream()), "rcv_fcn_forward")
        if self.check_finite:
            self.range_status(clear=True, raise_on_flag=True)
            if not (bool(torch.isfinite(seg).all()) and bool(torch.isfinite(radial).all())):
                raise FloatingPointError("non-finite FCN output in %s mode -- use dtype='fp32' for this checkpoint / input" % self.dtype_name)
        return seg, radial

    def range_status(self, clear=True, raise_on_flag=False):
        """Waits for the current stream and returns the engine's range-flag word (0 = every stored activation stayed inside the
        half range; bit 0: an f16x3 pair was clamped at 65504, bit 1: an fp16 store overflowed to infinity); fp32 / bf16 engines
        always return 0.  raise_on_flag: raise _hip.HalfRangeError (a FloatingPointError) instead."""
        flags = ctypes.c_uint(0)
        rc = self.lib.rcv_fcn_range_status(self.handle, ctypes.byref(flags), int(bool(clear)), _hip.current_stream())
        if rc not in (_hip.RCV_OK, _hip.RCV_E_RANGE):
            _hip.check(rc, "rcv_fcn_range_status")
        if flags.value and raise_on_flag:
            raise _hip.HalfRangeError("an activation left the half-precision range (|v| > 65504) in %s mode: the maps of that forward are "
                                      "clipped, not the reference's fp32 values (flags %d) -- use dtype='fp32' for this checkpoint"
                                      % (self.dtype_name, flags.value))
        return int(flags.value)

    def capture(self, x, out=None):
        """One forward for fixed buffers as a HIP graph (the ~170 launches of a forward are launch-bound at small batches).

        x: the static input tensor (refill it in place before a replay); returns (graph, seg, radial): `graph.replay()`
        re-runs the captured launches on the current stream and leaves the result in seg / radial.  Per-op profiling must
        be off; the layer program, options and batch size are frozen into the graph."""
        import torch

        B = x.shape[0]
        self.workspace(B)
        if out is None:
            out = (torch.empty((self.groups, B, 1, self.H, self.W), dtype=torch.float32, device=x.device),
                   torch.empty((self.groups, B, 1, self.H, self.W), dtype=torch.float32, device=x.device))
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up outside the capture (kernel attributes are registered on first use)
            self.forward(x, out=out)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.forward(x, out=out)
        return graph, out[0], out[1]

    def debug_stage(self, B, g, name):
        """Activation of a named stage as float32 [B,C,H,W] (engine must be built with reuse_slots=False)."""
        t, shape = self.stages[name]
        return self.debug_slot(B, g, t, shape).permute(0, 3, 1, 2).contiguous()

    def debug_slot(self, B, g, slot, shape_hwc):
        import torch

        h, w, c = shape_hwc
        out = torch.empty((B, h, w, c), dtype=torch.float32, device=self.device)
        _hip.check(self.lib.rcv_fcn_debug_slot(self.handle, _hip.ptr(self._ws), B, g, slot, _hip.ptr(out),
                                               _hip.current_stream()), "rcv_fcn_debug_slot")
        return out

    def set_option(self, key, value):
        _hip.check(self.lib.rcv_fcn_set_option(self.handle, int(key), int(value)))
        if int(key) == 8:
            self.radial_f16 = bool(value)

    def set_profiling(self, enable):
        _hip.check(self.lib.rcv_fcn_set_profiling(self.handle, int(bool(enable))))

    def get_profile(self):
        c, o, f = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _hip.check(self.lib.rcv_fcn_get_profile(self.handle, ctypes.byref(c), ctypes.byref(o), ctypes.byref(f)))
        return dict(conv_ms=c.value, other_ms=o.value, conv_flop_per_image_group=f.value)

    def get_op_times(self):
        arr = (ctypes.c_double * len(self.ops))()
        _hip.check(self.lib.rcv_fcn_get_op_times(self.handle, arr, len(self.ops)))
        return list(arr)


def preprocess_rgb(rgb_u8):
    """uint8 [B,H,W,3] CUDA tensor -> normalised float32 [B,3,H,W] (reference AccumulatorSpace.py:140-151)."""
    import torch

    L = _hip.lib()
    if rgb_u8.dim() == 3:
        rgb_u8 = rgb_u8.unsqueeze(0)
    rgb_u8 = rgb_u8.contiguous()
    B, H, W, _ = rgb_u8.shape
    x = torch.empty((B, 3, H, W), dtype=torch.float32, device=rgb_u8.device)
    _hip.check(L.rcv_preprocess_rgb(_hip.ptr(rgb_u8), B, H, W, _hip.ptr(x), _hip.current_stream()), "rcv_preprocess_rgb")
    return x
