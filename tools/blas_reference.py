#!/usr/bin/env python3
"""Developer reference point (GPU box): what does the vendor BLAS behind torch.mm reach in fp32 on the GEMM shapes of
cfg2 / cfg3, next to this repo's gemm_kernel (avsep_op_linear)?  Not used by the product path."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
from av_separation import _native
lib = _native.load()
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
print(f"{'M x N x K':>22s} {'torch.mm us':>12s} {'TF':>7s} {'this repo us':>13s} {'TF':>7s}")
for M, N, K in ((2016, 256, 256), (2016, 768, 256), (2016, 1024, 256), (2016, 256, 1024), (16064, 512, 512), (16064, 1536, 512),
                (16064, 2048, 512), (16064, 512, 2048), (16032, 2048, 512)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); y = torch.empty(M, N, device=dev)
    wt = w.t().contiguous()
    t_blas = min(timeit(lambda: torch.mm(x, w.t(), out=y)), timeit(lambda: torch.mm(x, wt, out=y)))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    t_mine = timeit(lambda: lib.avsep_op_linear(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), M, N, K, 0, st))
    fl = 2.0 * M * N * K
    print(f"{M:>8d} x{N:>5d} x{K:>5d} {t_blas*1e6:12.1f} {fl/t_blas/1e12:7.1f} {t_mine*1e6:13.1f} {fl/t_mine/1e12:7.1f}", flush=True)
