"""TEST INFRASTRUCTURE (checker only; nothing under rcvpose_amd/ imports this).

numpy restatement of the q6 side band of the split-precision FCN engine (include/rcvpose_hip.h, engine option 11): the e2m3 block
format of `v_mfma_scale_f32_16x16x128_f8f6f4` and the arithmetic of a convolution whose cross terms run in it.  There is no
reference counterpart (the reference's forward is fp32, models/fcnresnet.py:121-191): this pins the ENGINE's storage format and
kernel arithmetic; the engine's accuracy claim is tested against the reference's golden forward elsewhere
(tests/test_gpu_fcn_x3.py).

Format (per 16 consecutive channels = one block): scale 2^(floor(log2 max|v|) - 2) as an e8m0 byte (biased by 127; all-zero block:
byte 1), 16 codes sign | exponent (2 bits, bias 1) | mantissa (3 bits) of v / scale, rounded to nearest even, saturating at 7.5,
element j in bits [6j, 6j+6) of 12 bytes.  Hardware behaviour measured with tools/probe/fp6_mfma_probe.hip.
"""
import numpy as np

E2M3 = np.array([(m / 8.0 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 1)) for e in range(4) for m in range(8)])  # magnitudes by code & 31


def split_halves(v):
    """float64 -> (hi, lo) float16 with hi = half(v), lo = half(v - hi) (the engine's storage of every value)."""
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float64)).astype(np.float16)
    return hi, lo


def block_scale_byte(blk):
    """[..., 16] values -> e8m0 byte of the block scale."""
    amax = np.abs(blk.astype(np.float64)).max(axis=-1)
    _, e = np.frexp(amax)  # amax = m 2^e, m in [0.5, 1): floor(log2 amax) = e - 1
    biased = np.where(amax > 0, e - 1 + 127, 0)
    return (np.maximum(biased, 3) - 2).astype(np.int64)


def encode(blk):
    """[..., 16] values -> (codes uint8 [..., 16], scale byte int64 [...])."""
    sb = block_scale_byte(blk)
    y = blk.astype(np.float64) / np.exp2(sb - 127.0)[..., None]
    a = np.minimum(np.abs(y), 7.5)
    # nearest representable magnitude, ties to the even mantissa
    d = np.abs(a[..., None] - E2M3)                         # [..., 16, 32]
    best = d.min(axis=-1, keepdims=True)
    cand = d == best
    even = (np.arange(32) & 1) == 0
    pick = np.where((cand & even).any(axis=-1), (cand & even).argmax(axis=-1), cand.argmax(axis=-1))
    return (pick | (np.signbit(blk).astype(np.int64) << 5)).astype(np.uint8), sb


def decode(codes, sb):
    """-> float64 values the matrix instruction multiplies."""
    c = codes.astype(np.int64)
    mag = E2M3[c & 31]
    return np.where((c >> 5) & 1, -mag, mag) * np.exp2(sb.astype(np.float64) - 127.0)[..., None]


def quantize(v):
    """[..., C] (C % 16 == 0) values -> the values their q6 blocks stand for."""
    shp = v.shape
    b = v.reshape(*shp[:-1], shp[-1] // 16, 16)
    c, s = encode(b)
    return decode(c, s).reshape(shp)


def chunk_bytes(codes, sb, scale_byte):
    """16 codes + scale -> the 16-byte chunk (scale_byte 12: activations, 14: weights)."""
    c = codes.astype(np.uint64)
    out = np.zeros(codes.shape[:-1] + (16,), dtype=np.uint8)
    bits = np.zeros(codes.shape[:-1], dtype=object)
    for j in range(16):
        bits = bits + (c[..., j].astype(object) << (6 * j))
    for k in range(12):
        out[..., k] = ((bits >> (8 * k)) & 255).astype(np.uint8)
    out[..., scale_byte] = sb.astype(np.uint8)
    return out


def unpack_chunk(chunk, scale_byte):
    """16-byte chunks [..., 16] uint8 -> (codes [..., 16], scale byte [...])."""
    bits = np.zeros(chunk.shape[:-1], dtype=object)
    for k in range(12):
        bits = bits + (chunk[..., k].astype(object) << (8 * k))
    codes = np.stack([((bits >> (6 * j)) & 63).astype(np.uint8) for j in range(16)], axis=-1)
    return codes, chunk[..., scale_byte].astype(np.int64)


def activation_rows(hi, lo):
    """hi, lo float16 [..., 32] (one 32-channel group per row) -> the q6 row's second half, uint8 [..., 64]:
    chunks hi[0:16], hi[16:32], lo[0:16], lo[16:32], scale in byte 12."""
    parts = []
    for src in (hi, lo):
        for half in (0, 1):
            c, s = encode(src[..., 16 * half:16 * half + 16].astype(np.float64))
            parts.append(chunk_bytes(c, s, 12))
    return np.concatenate(parts, axis=-1)


def conv3x3_q6(x, w, bias, relu=True):
    """One 3x3 / stride-1 / pad-1 convolution as the engine computes it under option 11, float64 accumulation.
    x: float64 [B,H,W,C] (22-bit valued: hi + lo), w: float64 [Cout,3,3,C] weights AS STORED (scaled per output channel, 22-bit
    valued), bias / post-scale applied by the caller.  y = hi*hi + q(a_hi) q(w_lo) + q(a_lo) q(w_hi)."""
    a_hi, a_lo = split_halves(x)
    w_hi, w_lo = split_halves(w)
    ah, al = a_hi.astype(np.float64), a_lo.astype(np.float64)
    wh, wl = w_hi.astype(np.float64), w_lo.astype(np.float64)
    qah, qal, qwh, qwl = quantize(ah), quantize(al), quantize(wh), quantize(wl)
    B, H, W, C = x.shape
    y = np.zeros((B, H, W, w.shape[0]))

    def pad(t):
        return np.pad(t, ((0, 0), (1, 1), (1, 1), (0, 0)))

    pah, pqah, pqal = pad(ah), pad(qah), pad(qal)
    for r in range(3):
        for s in range(3):
            y += pah[:, r:r + H, s:s + W] @ wh[:, r, s].T
            y += pqah[:, r:r + H, s:s + W] @ qwl[:, r, s].T
            y += pqal[:, r:r + H, s:s + W] @ qwh[:, r, s].T
    return y
