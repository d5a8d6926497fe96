#!/usr/bin/env python3
"""Developer tool (GPU box): run avsep_op_linear for ONE shape N times (target of rocprofv3 --pmc passes)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
from av_separation import _native
M, N, K = (int(v) for v in sys.argv[1:4]); reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
lib = _native.load(); dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(reps):
    lib.avsep_op_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), M, N, K, 1, st)
torch.cuda.synchronize()
