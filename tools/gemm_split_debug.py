import os
os.environ["AVSEP_LIB"] = "dev"; os.environ["AVSEP_GEMM_SPLIT"] = "1"
import ctypes as C, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
from av_separation import _native
lib = _native.load(); dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
M, N, K = 128, 128, 32
def run(x, w):
    y = torch.empty(M, N, device=dev)
    assert lib.avsep_op_linear(x.contiguous().data_ptr(), w.contiguous().data_ptr(), None, None, y.data_ptr(), M, N, K, 0, st) == 0
    torch.cuda.synchronize()
    return y
def tr(t):
    return (t.view(torch.int32) & -65536).view(torch.float32)
x = torch.randn(M, K, device=dev)
hi = tr(x); mid = tr(x - hi); lo = tr(x - hi - mid)
for k in (0, 5):
    w = torch.zeros(N, K, device=dev); w[:, k] = 1.0
    y = run(x, w)
    for m in (0, 1, 17, 100):
        print(f"k={k} m={m}: x {float(x[m,k])!r} y {float(y[m,0])!r} | hi {float(hi[m,k])!r} hi+mid {float(hi[m,k]+mid[m,k])!r} | y-hi {float(y[m,0]-hi[m,k]):.3e} mid {float(mid[m,k]):.3e} lo {float(lo[m,k]):.3e}")
# x nonzero only in column k (all rows), w one-hot
xk = torch.zeros(M, K, device=dev); xk[:, 0] = x[:, 0]
w = torch.zeros(N, K, device=dev); w[:, 0] = 1.0
y = run(xk, w)
print("x only column 0, w one-hot 0: max err", float((y - x[:, 0:1]).abs().max()))
# x nonzero in columns 0 and 1 (same packed dword), w one-hot 0
xk[:, 1] = x[:, 1]
y = run(xk, w)
print("x columns 0,1, w one-hot 0: max err", float((y - x[:, 0:1]).abs().max()))
xk[:, 1] = 0; xk[:, 2] = x[:, 2]
y = run(xk, w)
print("x columns 0,2, w one-hot 0: max err", float((y - x[:, 0:1]).abs().max()))
xk[:, 2] = 0; xk[:, 9] = x[:, 9]
y = run(xk, w)
print("x columns 0,9, w one-hot 0: max err", float((y - x[:, 0:1]).abs().max()))
