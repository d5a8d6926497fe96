#!/usr/bin/env python3
"""Developer tool (GPU box): the two-term fp16 attention against the three-term bf16 one and the fp32-MFMA kernel: time and error."""
import os, sys, time, math, ctypes as C
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
torch.manual_seed(0)
print("(B, h, L): fp32 us | bf16x3 us TF | fp16x2 us TF | max err vs float64: fp32, bf16x3, fp16x2")
for B, h, L in ((64, 8, 251), (32, 8, 501), (32, 8, 251), (16, 8, 501), (1, 8, 501)):
    d = 64 * h
    qkv = torch.randn(B, L, 3 * d, device=dev)
    o = [torch.empty(B, L, d, device=dev) for _ in range(3)]
    ex = lambda a: 14 - math.frexp(float(a.abs().max()) * (1 + 1e-6))[1] - 6
    eq, ek, ev = ex(qkv[..., :d]), ex(qkv[..., d:2 * d]), ex(qkv[..., 2 * d:])
    qp, kp, vp = qkv.data_ptr(), qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d
    f0 = lambda: lib.avsep_op_attention(qp, 3 * d, kp, 3 * d, vp, 3 * d, o[0].data_ptr(), d, B, h, 64, L, L, st)
    f1 = lambda: lib.avsep_op_attention_split(qp, 3 * d, kp, 3 * d, vp, 3 * d, o[1].data_ptr(), d, B, h, 64, L, L, st)
    f2 = lambda: lib.avsep_op_attention_h2(qp, 3 * d, kp, 3 * d, vp, 3 * d, o[2].data_ptr(), d, B, h, 64, L, L, eq, ek, ev, st)
    assert f0() == 0 and f1() == 0 and f2() == 0, lib.avsep_last_error()
    ts = [timeit(f) for f in (f0, f1, f2)]
    nb = min(2, B)
    q64 = qkv[:nb, :, :d].double().reshape(nb, L, h, 64).transpose(1, 2)
    k64 = qkv[:nb, :, d:2 * d].double().reshape(nb, L, h, 64).transpose(1, 2); v64 = qkv[:nb, :, 2 * d:].double().reshape(nb, L, h, 64).transpose(1, 2)
    ref = (torch.softmax(q64 @ k64.transpose(-1, -2), -1) @ v64).transpose(1, 2).reshape(nb, L, d)
    errs = [float((x[:nb].double() - ref).abs().max()) for x in o]
    fl = 4.0 * B * h * L * L * 64
    print((B, h, L), f"{ts[0] * 1e6:7.1f} | {ts[1] * 1e6:7.1f} {fl / ts[1] / 1e12:6.1f} | {ts[2] * 1e6:7.1f} {fl / ts[2] / 1e12:6.1f} | "
          f"{errs[0]:.2e} {errs[1]:.2e} {errs[2]:.2e}")
