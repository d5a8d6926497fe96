#!/usr/bin/env python3
"""Developer tool (GPU box): per-kernel cost of DEPENDENT kernel chains replayed from a hipGraph --
separates the launch-boundary cost from in-kernel fixed cost for the small kernels of the forward."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
from av_separation import _native
lib = _native.load(); dev = torch.device("cuda:0")
def st(): return C.c_void_p(torch.cuda.current_stream().cuda_stream)
def bench(name, fn, n=100, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): g.replay()
        e1.record(); torch.cuda.synchronize()
    print(f"{name:44s} {e0.elapsed_time(e1) / reps / n * 1e3:8.2f} us per kernel")
x1 = torch.randn(1, 2, 64, device=dev); y1 = torch.empty(1, 4, 64, device=dev)
bench("tiny interp (1 workgroup)", lambda: lib.avsep_op_interp_linear(x1.data_ptr(), y1.data_ptr(), 1, 2, 4, 64, st()))
for M, d in ((2016, 256), (16064, 512)):
    x = torch.randn(M, d, device=dev); g_ = torch.ones(d, device=dev); b_ = torch.zeros(d, device=dev); y = torch.empty_like(x)
    bench(f"layernorm {M}x{d} (x->y, same buffers)", lambda: lib.avsep_op_layernorm(x.data_ptr(), g_.data_ptr(), b_.data_ptr(), y.data_ptr(), M, d, 1e-5, st()))
    z = torch.empty_like(x)
    def pingpong():
        lib.avsep_op_layernorm(x.data_ptr(), g_.data_ptr(), b_.data_ptr(), y.data_ptr(), M, d, 1e-5, st())
        lib.avsep_op_layernorm(y.data_ptr(), g_.data_ptr(), b_.data_ptr(), x.data_ptr(), M, d, 1e-5, st())
    bench(f"layernorm {M}x{d} dependent ping-pong (per 2)", pingpong, n=50)
for (M, N, K) in ((2016, 256, 32), (2016, 256, 256), (2016, 256, 1024), (2016, 1024, 256), (2016, 1024, 32), (63, 256, 256)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.zeros(N, device=dev); y = torch.empty(M, N, device=dev)
    bench(f"linear {M}x{N}x{K}", lambda: lib.avsep_op_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), M, N, K, 0, st()))
B, h, dh, L = 32, 4, 64, 63
q = torch.randn(B, L, 3 * h * dh, device=dev); o = torch.empty(B, L, h * dh, device=dev)
bench("attention B32 h4 L63", lambda: lib.avsep_op_attention(q.data_ptr(), 768, q.data_ptr() + 1024, 768, q.data_ptr() + 2048, 768, o.data_ptr(), 256, B, h, dh, L, L, st()))
