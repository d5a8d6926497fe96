#!/usr/bin/env python3
"""Developer tool (GPU box): what does the LAST, partially filled round of tiles cost the large GEMM?  (stream-K
on the last partial round.)  128x64 tiles, 3 workgroups per CU = 768 resident slots: the model's M = 16064 rows give 5.25 rounds at
N = 2048 and 1.31 at N = 512.  Time the same (N, K) at row counts that make exactly full rounds and at the model's: if TFLOP/s is flat
across them, the partial round costs nothing a split of its tiles could win back."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
from av_separation import _native
lib = _native.load(); dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, n=40):
    for _ in range(8): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n


for N, K in ((2048, 512), (1536, 512), (512, 512), (512, 2048)):
    nbn = N // 64
    print(f"N={N} K={K}: column tiles {nbn}")
    rows = sorted({16064} | {128 * (768 * r // nbn) for r in (1, 2, 3, 4, 5, 6) if 768 * r // nbn > 0} | {128 * ((768 * r + 384) // nbn) for r in (1, 2, 5)})
    for M in rows:
        if M < 1024 or M > 40000: continue
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; y = torch.empty(M, N, device=dev)
        f = lambda: lib.avsep_op_linear(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), M, N, K, 0, st)
        t = timeit(f)
        tiles = ((M + 127) // 128) * nbn
        print(f"   M={M:6d}  tiles {tiles:5d} = {tiles / 768:5.2f} rounds   {t * 1e6:8.1f} us   {2.0 * M * N * K / t / 1e12:6.1f} TFLOP/s")
