#!/usr/bin/env python3
"""Developer tool (GPU box): error statistics of the fp32-MFMA GEMM, the split-precision GEMM and torch's fp32 matmul against
float64 on LayerNorm-like activations x a trained-like weight: max and RMS error, mean (signed) error relative to |y|, and the
number of outputs whose SIGN differs from float64's (what a ReLU behind the GEMM turns into gradient differences)."""
import os
os.environ["AVSEP_LIB"] = "dev"
import ctypes as C, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
from av_separation import _native
lib = _native.load(); dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(1)
print("shape                    kernel        max|e|/max|y|   rms e / rms y   mean(e*sign(y))/rms y   sign flips vs float64")
for M, N, K in ((4016, 2048, 512), (4016, 512, 2048), (16064, 2048, 512)):
    x = torch.nn.functional.layer_norm(torch.randn(M, K, device=dev) * 3 + 1, (K,)) * (1 + 0.1 * torch.randn(K, device=dev)) + 0.05 * torch.randn(K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = 0.02 * torch.randn(N, device=dev)
    ref = x.double() @ w.double().t() + b.double()
    ys = {}
    y = torch.empty(M, N, device=dev)
    assert lib.avsep_op_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), M, N, K, 0, st) == 0
    ys["fp32 MFMA GEMM"] = y.clone()
    for v in ("1", "2", "3"):
        os.environ["AVSEP_SPLIT_VARIANT"] = v
        assert lib.avsep_op_linear_split(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), M, N, K, 0, st) == 0
        ys["split, kernel " + v] = y.clone()
    os.environ.pop("AVSEP_SPLIT_VARIANT", None)
    ys["torch fp32 (addmm)"] = torch.addmm(b, x, w.t())
    # the split scheme emulated in float64 (no accumulation error): what the dropped terms alone cost
    def cut(t):
        u = t.view(torch.int32)
        hi = (u & -65536).view(torch.float32); r1 = t - hi
        mid = (r1.view(torch.int32) & -65536).view(torch.float32); r2 = r1 - mid
        lo = (r2.view(torch.int32) & -65536).view(torch.float32)
        return hi.double(), mid.double(), lo.double()
    xh, xm, xl = cut(x); wh, wm, wl = cut(w)
    emu = xh @ wh.t() + xh @ wm.t() + xm @ wh.t() + xm @ wm.t() + xh @ wl.t() + xl @ wh.t() + b.double()
    ys["six products in float64"] = emu
    for k, yv in ys.items():
        e = yv.double() - ref
        print(f"({M:6d},{N:5d},{K:5d})  {k:24s} {float(e.abs().max() / ref.abs().max()):.3e}      {float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.3e}      "
              f"{float((e * ref.sign()).mean() / ref.pow(2).mean().sqrt()):+.3e}            {int((yv.double().sign() != ref.sign()).sum())}", flush=True)
