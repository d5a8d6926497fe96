#!/usr/bin/env python3
"""Developer tool (GPU box): launch ONE kernel of the two-term path a few times (for PMC passes: tools/pmc_kernel.sh).
    one_kernel.py gemm_h2 M N K [iters]          avsep_op_linear_h2, fp32 output, bias
    one_kernel.py attn_h2 B h L [iters]          avsep_op_attention_h2 (dh = 64, self-attention shape)
    one_kernel.py attn_split B h L [iters]       avsep_op_attention_split (three bf16 terms)"""
import os, sys, math, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
from av_separation import _native
lib = _native.load(); dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
kind = sys.argv[1]; a, b, c = (int(x) for x in sys.argv[2:5]); it = int(sys.argv[5]) if len(sys.argv) > 5 else 10
torch.manual_seed(0)
if kind == "gemm_h2":
    M, N, K = a, b, c
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; bias = torch.randn(N, device=dev)
    xp = torch.zeros(K // 32 * 2 * M * 32, dtype=torch.int16, device=dev)
    assert lib.avsep_op_split_h2(x.data_ptr(), K, xp.data_ptr(), M, M, K, None, 10, st) == 0
    ew = torch.zeros(N, dtype=torch.int32, device=dev); l2 = torch.zeros(N, device=dev)
    assert lib.avsep_op_h2_row_stats(w.data_ptr(), N, K, ew.data_ptr(), l2.data_ptr(), st) == 0
    wp = torch.zeros(K // 32 * 2 * N * 32, dtype=torch.int16, device=dev)
    assert lib.avsep_op_split_h2(w.data_ptr(), K, wp.data_ptr(), N, N, K, ew.data_ptr(), 0, st) == 0
    cs = torch.ldexp(torch.ones(N, device=dev), -(ew + 10)); y = torch.empty(M, N, device=dev)
    f = lambda: lib.avsep_op_linear_h2(xp.data_ptr(), M, wp.data_ptr(), N, cs.data_ptr(), None, bias.data_ptr(), None, y.data_ptr(), None, 0, 0, M, N, K, 0, st)
else:
    B, h, L = a, b, c; d = 64 * h
    qkv = torch.randn(B, L, 3 * d, device=dev); o = torch.empty(B, L, d, device=dev)
    ex = lambda t: 14 - math.frexp(float(t.abs().max()) * 1.000001)[1] - 6
    eq, ek, ev = ex(qkv[..., :d]), ex(qkv[..., d:2 * d]), ex(qkv[..., 2 * d:])
    qp, kp, vp = qkv.data_ptr(), qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d
    if kind == "attn_h2":
        f = lambda: lib.avsep_op_attention_h2(qp, 3 * d, kp, 3 * d, vp, 3 * d, o.data_ptr(), d, B, h, 64, L, L, eq, ek, ev, st)
    else:
        f = lambda: lib.avsep_op_attention_split(qp, 3 * d, kp, 3 * d, vp, 3 * d, o.data_ptr(), d, B, h, 64, L, L, st)
for _ in range(it):
    assert f() == 0, lib.avsep_last_error()
torch.cuda.synchronize()
