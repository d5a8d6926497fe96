#!/usr/bin/env python3
"""Developer tool (GPU box): time the stand-alone LayerNorm launch on the row-tensor shapes of the workloads and
print the achieved HBM rate next to a device copy of the same bytes."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
from av_separation import _native
lib = _native.load()
dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, d) in [(16064, 512), (16064, 256), (32128, 512), (2016, 256), (4032, 256), (64256, 512)]:
    x = torch.randn(M, d, device=dev); y = torch.empty_like(x); g = torch.ones(d, device=dev); b = torch.zeros(d, device=dev)
    def ln(): lib.avsep_op_layernorm(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), M, d, 1e-5, st)
    def cp(): y.copy_(x)
    res = []
    for f in (ln, cp):
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 200
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n * 1e3)
    byt = 2.0 * M * d * 4
    print(f"M={M:6d} d={d:4d}  layernorm {res[0]:7.2f} us = {byt/res[0]/1e6:6.2f} TB/s   copy {res[1]:7.2f} us = {byt/res[1]/1e6:6.2f} TB/s")
