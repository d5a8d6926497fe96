#!/usr/bin/env python3
"""Developer tool: time avsep_op_linear on the large GEMM shapes under the tile chosen by AVSEP_GEMM_TILE (or auto)."""
import os
os.environ.setdefault("AVSEP_LIB", "dev")   # developer switches live in libavsep_hip_dev.so only
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
from av_separation import _native
lib = _native.load(); dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
out = []
for M, N, K in ((16064, 512, 512), (16064, 1536, 512), (16064, 2048, 512), (16064, 512, 2048), (4016, 2048, 512), (4016, 512, 2048)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); y = torch.empty(M, N, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.avsep_op_linear(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), M, N, K, 0, st)
    if rc != 0:
        out.append("   n/a"); continue
    t = timeit(lambda: lib.avsep_op_linear(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), M, N, K, 0, st))
    ok = float((y - x @ w.t()).abs().max()) < 1e-2
    out.append(f"{t*1e6:7.1f}us {2.0*M*N*K/t/1e12:5.1f}TF{'' if ok else ' WRONG'}")
print(os.environ.get("AVSEP_GEMM_TILE", "auto"), " | ".join(out))
