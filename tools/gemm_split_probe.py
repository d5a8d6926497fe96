#!/usr/bin/env python3
"""Developer tool (GPU box): the split-precision GEMM prototype (gemm_split.hip: fp32 operands as three bf16 terms, six bf16 MFMA
products, fp32 accumulation) against the shipped fp32-MFMA GEMM on the model's large shapes: time, and error against float64."""
import os
os.environ["AVSEP_LIB"] = "dev"
import ctypes as C, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
from av_separation import _native
lib = _native.load(); dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, n=30):
    for _ in range(6): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n


torch.manual_seed(0)
print("shape (M, N, K)            fp32 MFMA: us  TFLOP/s  err/max|y| | split 128x128: us  TFLOP/s  err | split 256x128: us  TFLOP/s  err | split 64x64: us  TFLOP/s  err | speed-ups")
for M, N, K, act, res in ((16064, 2048, 512, 1, False), (16064, 1536, 512, 0, False), (16064, 512, 512, 0, True), (16064, 512, 2048, 0, True),
                          (16032, 2048, 512, 1, False), (16032, 512, 2048, 0, True), (8192, 1024, 1024, 0, False),
                          (4016, 2048, 512, 0, False), (4016, 1536, 512, 0, False), (4016, 512, 512, 0, True), (4016, 512, 2048, 0, True),
                          (3200, 2048, 512, 1, False), (3200, 1536, 512, 0, False), (3200, 512, 512, 0, True), (3200, 512, 2048, 0, True), (2048, 512, 512, 0, False), (1004, 2048, 512, 1, False), (1004, 512, 2048, 0, True), (251, 1536, 512, 0, False), (251, 512, 2048, 0, True), (2016, 768, 256, 0, False), (777, 260, 96, 2, True)):
    x = (torch.randn(M, K, device=dev) * 2 + 0.7); w = torch.randn(N, K, device=dev) * 0.06; b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    ref = x.double() @ w.double().t() + b.double()
    ref = {0: ref, 1: torch.relu(ref), 2: torch.nn.functional.gelu(ref)}[act]
    if res: ref = ref + r.double()
    sc = float(ref.abs().max())
    fl = 2.0 * M * N * K
    out = []
    for variant in (None, "1", "2", "3"):
        y = torch.empty(M, N, device=dev)
        call = lambda: lib.avsep_op_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, y.data_ptr(), M, N, K, act, st)
        os.environ.pop("AVSEP_GEMM_SPLIT", None); os.environ.pop("AVSEP_SPLIT_VARIANT", None)
        if variant:
            os.environ["AVSEP_GEMM_SPLIT"] = "1"; os.environ["AVSEP_SPLIT_VARIANT"] = variant
        assert call() == 0
        t = timeit(call)
        out.append((t, float((y.double() - ref).abs().max()) / sc))
    os.environ.pop("AVSEP_GEMM_SPLIT", None); os.environ.pop("AVSEP_SPLIT_VARIANT", None)
    cells = " | ".join(f"{t * 1e6:8.1f} {fl / t / 1e12:7.1f}  {e:.2e}" for t, e in out)
    print(f"({M:6d},{N:5d},{K:5d}) act {act} res {int(res)}  {cells} | x{out[0][0] / out[1][0]:.2f} x{out[0][0] / out[2][0]:.2f} x{out[0][0] / out[3][0]:.2f}", flush=True)
