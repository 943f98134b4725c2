#!/usr/bin/env python3
"""A lower bound for Winograd F(2x2, 3x3) on the layer-3 shape (256 -> 256 at 30x40, 159 images per step), measured with kernels
of the tree: the 16 element-wise GEMMs of the transformed domain ARE sixteen 1x1 convolutions [tiles x 256] . [256 x 256] -- one
per position of the 4 x 4 transformed tile, tiles = 15 x 20 per image.  Timed here as ONE 1x1 convolution 256 -> 256 over
16 x 300 x 159 = 763 200 rows on the engine's GEMM kernels (split precision, the same three MFMAs per product), next to the direct
3x3 convolution of the same layer (k_conv_halo).  What the bound leaves out, all of it extra: the input transform B^T d B (reads
the activation, writes 4x its bytes), the output transform A^T m A (needs the 16 products in fp32: another 4x), and sixteen
different weight sets instead of one.

    python tools/probe/winograd_core_bound.py          (on the GPU box)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from rcvpose_amd import _hip, fcn_engine


def one_op(cin, cout, H, W, k, B):
    rng = np.random.default_rng(1)
    w = rng.standard_normal((cout, k, k, cin)) * (1.0 / np.sqrt(k * k * cin))
    flat, scale = fcn_engine.pack_split(torch.from_numpy(w), _hip.OP_CONV)
    wdev = flat.cuda()
    bdev = torch.cat([torch.zeros(cout, dtype=torch.float32), scale]).cuda()
    op = _hip.Op()
    for f, val in dict(kind=_hip.OP_CONV, src0=0, src1=-1, res=-1, dst=1, c0=cin, c1=0, cout=cout, hin=H, win=W, hout=H, wout=W, ksize=k,
                       stride=1, pad=k // 2, relu=1, w_off=0, b_off=0, wq_off=0).items():
        setattr(op, f, val)
    slots = (ctypes.c_int64 * 2)(H * W * cin, H * W * cout)
    h = ctypes.c_void_p()
    L = _hip.lib()
    _hip.check(L.rcv_fcn_create((_hip.Op * 1)(op), 1, 2, slots, 1, _hip.RCV_F16X3, _hip.ptr(wdev), _hip.ptr(bdev), ctypes.byref(h)), "rcv_fcn_create")
    nbytes = int(L.rcv_fcn_workspace_bytes(h, B))
    ws = (torch.rand(nbytes // 2, device="cuda") * 2).to(torch.float16).view(torch.uint8)   # (any finite halves)
    x = torch.zeros((B, 3, H, W), device="cuda")
    seg = torch.zeros((1, B, 1, H, W), device="cuda")
    rad = torch.zeros_like(seg)

    def run():
        _hip.check(L.rcv_fcn_forward(h, _hip.ptr(x), B, _hip.ptr(seg), _hip.ptr(rad), _hip.ptr(ws), nbytes, _hip.current_stream()), "rcv_fcn_forward")

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    L.rcv_fcn_range_status(h, None, 1, _hip.current_stream())
    L.rcv_fcn_destroy(h)
    del wdev, bdev, ws
    torch.cuda.empty_cache()
    return ms


if __name__ == "__main__":
    B = 159                      # 53 frames x 3 networks of a bench step
    direct = one_op(256, 256, 30, 40, 3, B)
    core = one_op(256, 256, 60, 80, 1, B)   # 159 x 4800 rows = 16 positions x 300 tiles x 159 images
    flop = 2.0 * B * 1200 * 256 * 256 * 9
    print("direct 3x3 256->256 at 30x40, %d images (k_conv_halo):           %.3f ms  (%.0f TFLOP/s algorithmic)" % (B, direct, flop / direct / 1e9))
    print("Winograd F(2x2,3x3) GEMM core alone, 16 x [%d x 256].[256 x 256]: %.3f ms  (16/36 of the multiplies)" % (B * 300, core))
    print("  bytes of that core: transformed input %d MB + products (fp32-grade) %d MB; the direct kernel moves %d MB" %
          (B * 4800 * 256 * 4 // 2**20, B * 4800 * 256 * 4 // 2**20, 2 * B * 1200 * 256 * 4 // 2**20))
