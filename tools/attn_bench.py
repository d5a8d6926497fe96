#!/usr/bin/env python3
"""Developer tool (GPU box): time avsep_op_attention alone on the sequence shapes of the workloads; prints us per
launch and TFLOP/s (4 B h Lq Lk dh flops).  Variants through the developer environment switches of attention.hip."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
from av_separation import _native
lib = _native.load(); dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (B, h, dh, L, tag) in [(64, 8, 64, 251, "cfg3"), (32, 8, 64, 501, "cfg5"), (16, 8, 64, 251, "cfg4 batch"), (32, 4, 64, 63, "cfg2 audio"), (32, 4, 64, 50, "cfg2 video")]:
    d = h * dh
    qkv = torch.randn(B * L, 3 * d, device=dev); o = torch.empty(B * L, d, device=dev)
    f = lambda: lib.avsep_op_attention(qkv.data_ptr(), 3 * d, qkv.data_ptr() + 4 * d, 3 * d, qkv.data_ptr() + 8 * d, 3 * d, o.data_ptr(), d, B, h, dh, L, L, st)
    for _ in range(5): rc = f()
    assert rc == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 100
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{tag:12s} B={B:3d} h={h} dh={dh} L={L:4d}: {us:8.2f} us  {4.0*B*h*L*L*dh/us/1e6:6.1f} TFLOP/s")
