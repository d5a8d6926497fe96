#!/usr/bin/env python3
"""Developer tool (GPU box): long-sequence attention alone against the batch size -- how much of its efficiency is the
workgroup count not being a multiple of the chip's resident slots (256 CUs x 3 workgroups)?"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
from av_separation import _native
lib = _native.load(); dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
h, dh = 8, 64
for L in (251, 501):
    for B in (16, 24, 32, 48, 64, 72, 96, 144, 192):
        d = h * dh
        qkv = torch.randn(B * L, 3 * d, device=dev); o = torch.empty(B * L, d, device=dev)
        f = lambda: lib.avsep_op_attention(qkv.data_ptr(), 3 * d, qkv.data_ptr() + 4 * d, 3 * d, qkv.data_ptr() + 8 * d, 3 * d, o.data_ptr(), d, B, h, dh, L, L, st)
        for _ in range(5): rc = f()
        assert rc == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        nq = (L + 15) // 16; wgs = B * h * ((((nq + 1) // 2) + 3) // 4)
        print(f"L={L:4d} B={B:4d}: workgroups {wgs:5d} = {wgs/768:5.2f} x 768   {us:8.2f} us  {4.0*B*h*L*L*dh/us/1e6:6.1f} TFLOP/s", flush=True)
