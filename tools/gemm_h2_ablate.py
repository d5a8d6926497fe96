#!/usr/bin/env python3
"""Developer tool (GPU box): what bounds gemm_h2_kernel<128>?  Timing-only ablations (AVSEP_H2_ABL, dev library): 1 = half the LDS
fragment reads, 2 = no DMA in the loop, 3 = both, 4 = no MFMA, 6 = neither MFMA nor DMA (LDS reads alone), 8 = no epilogue, 10 = no epilogue and no DMA, 12 = no epilogue and no MFMA."""
import os
os.environ.setdefault("AVSEP_LIB", "dev")
import ctypes as C, sys, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
from av_separation import _native
lib = _native.load(); dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for M, N, K in ((16064, 2048, 512), (16064, 512, 2048), (16064, 1536, 512), (16064, 512, 512)):
    xp = torch.zeros(K // 32 * 2 * M * 32, dtype=torch.int16, device=dev); wp = torch.zeros(K // 32 * 2 * N * 32, dtype=torch.int16, device=dev)
    xp.random_(0, 1 << 14); wp.random_(0, 1 << 14)
    cs = torch.ones(N, device=dev); b = torch.zeros(N, device=dev); y = torch.empty(M, N, device=dev)
    row = []
    for abl in ("", "1", "2", "4", "8", "10", "12"):
        os.environ.pop("AVSEP_H2_ABL", None)
        if abl: os.environ["AVSEP_H2_ABL"] = abl
        f = lambda: lib.avsep_op_linear_h2(xp.data_ptr(), M, wp.data_ptr(), N, cs.data_ptr(), None, b.data_ptr(), None, y.data_ptr(), None, 0, 0, M, N, K, 0, st)
        assert f() == 0
        t = timeit(f)
        row.append(f"abl {abl or '0'}: {t * 1e6:6.1f} us {2.0 * M * N * K / t / 1e12:6.1f} TF")
    print((M, N, K), " | ".join(row))
