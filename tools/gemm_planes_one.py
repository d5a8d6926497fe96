#!/usr/bin/env python3
"""Developer tool (GPU box): run avsep_op_linear_planes for ONE shape N times (target of rocprofv3 --pmc passes).
   gemm_planes_one.py M N K [reps] ; AVSEP_PLANES_V picks the developer variant."""
import os
os.environ.setdefault("AVSEP_LIB", "dev")
import ctypes as C, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
from av_separation import _native
M, N, K = (int(v) for v in sys.argv[1:4]); reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
lib = _native.load(); dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.randn(M, K, device=dev) * 2 + 0.7; w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
xp = torch.zeros(K // 32 * 3 * M * 32, dtype=torch.int16, device=dev); wp = torch.zeros(K // 32 * 3 * N * 32, dtype=torch.int16, device=dev)
assert lib.avsep_op_split_planes(x.data_ptr(), K, xp.data_ptr(), M, M, K, st) == 0
assert lib.avsep_op_split_planes(w.data_ptr(), K, wp.data_ptr(), N, N, K, st) == 0
y = torch.empty(M, N, device=dev)
for _ in range(reps):
    assert lib.avsep_op_linear_planes(xp.data_ptr(), M, wp.data_ptr(), N, b.data_ptr(), None, y.data_ptr(), None, 0, M, N, K, 1, st) == 0
torch.cuda.synchronize()
