import os, sys, time, ctypes as C
sys.path.insert(0, "av-separation-transformer_amd")
import torch
from av_separation import _native
lib = _native.load(); dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for M, N, K in ((16064, 2048, 512), (8032, 2048, 512), (16064, 1024, 512)):
    xp = torch.zeros(K // 32 * 2 * M * 32, dtype=torch.int16, device=dev).random_(0, 1 << 13)
    wp = torch.zeros(K // 32 * 2 * N * 32, dtype=torch.int16, device=dev).random_(0, 1 << 13)
    cs = torch.full((N,), 1e-6, device=dev); b = torch.zeros(N, device=dev)
    yp = torch.zeros(N // 32 * 2 * M * 32, dtype=torch.int16, device=dev); y = torch.empty(M, N, device=dev)
    row = []
    for act in (0, 1, 2):
        f = lambda: lib.avsep_op_linear_h2(xp.data_ptr(), M, wp.data_ptr(), N, cs.data_ptr(), None, b.data_ptr(), None, None, yp.data_ptr(), M, 3, M, N, K, act, st)
        assert f() == 0, lib.avsep_last_error()
        row.append(f"act {act} planes out {timeit(f) * 1e6:6.1f} us")
    f = lambda: lib.avsep_op_linear_h2(xp.data_ptr(), M, wp.data_ptr(), N, cs.data_ptr(), None, b.data_ptr(), None, y.data_ptr(), None, 0, 0, M, N, K, 0, st)
    assert f() == 0
    row.append(f"fp32 out {timeit(f) * 1e6:6.1f} us")
    print((M, N, K), " | ".join(row))
