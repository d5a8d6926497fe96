#!/usr/bin/env python3
"""Developer tool (GPU box): run avsep_op_ln_linear (form 1: LayerNorm inside the GEMM) for ONE shape N times -- the target
of AVSEP_GEMM_DBG phase stamps and of rocprofv3 --pmc passes.   tools/gemm_ln_one.py M N K [reps]"""
import os
os.environ.setdefault("AVSEP_LIB", "dev")
import ctypes as C, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
from av_separation import _native
M, N, K = (int(v) for v in sys.argv[1:4]); reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
lib = _native.load(); dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
g = torch.rand(K, device=dev) + 0.5; be = torch.randn(K, device=dev)
y = torch.empty(M, N, device=dev); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(reps):
    rc = lib.avsep_op_ln_linear(x.data_ptr(), g.data_ptr(), be.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, M, N, K, 1, 1e-5, 1, st)
    assert rc == 0, lib.avsep_last_error()
torch.cuda.synchronize()
