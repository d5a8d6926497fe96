#!/usr/bin/env python3
"""Developer tool (GPU box): what would launching the audio and the visual instance of an encoder-layer GEMM as ONE kernel
buy?  Times the plain / LayerNorm-fused GEMM and the attention kernel at M = 2016 (audio), 1600 (visual) and 3616 (both)
rows for each tile override; one subprocess per override (the switches are read once per process)."""
import os
os.environ.setdefault("AVSEP_LIB", "dev")   # developer switches live in libavsep_hip_dev.so only
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
MS = (1600, 2016, 3616)
NK = ((768, 256), (256, 256), (1024, 256), (256, 1024))


def child():
    import torch
    from av_separation import _native
    lib = _native.load(); dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}

    def timeit(fn, n=60):
        for _ in range(8): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    for (N, K) in NK:
        for M in MS:
            x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
            y = torch.empty(M, N, device=dev); g = torch.randn(K, device=dev); be = torch.randn(K, device=dev)
            r = torch.randn(M, N, device=dev)
            out[f"plain {M}x{N}x{K}"] = timeit(lambda: lib.avsep_op_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr(), y.data_ptr(), M, N, K, 0, st))
            if K == 256:
                out[f"ln    {M}x{N}x{K}"] = timeit(lambda: lib.avsep_op_ln_linear(x.data_ptr(), g.data_ptr(), be.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, M, N, K, 1, C.c_float(1e-5), 1, st))
    for (B, L) in ((32, 63), (32, 50), (64, 63)):
        qkv = torch.randn(B * L, 768, device=dev); o = torch.empty(B * L, 256, device=dev)
        out[f"attn B{B} L{L}"] = timeit(lambda: lib.avsep_op_attention(qkv.data_ptr(), 768, qkv.data_ptr() + 1024, 768, qkv.data_ptr() + 2048, 768, o.data_ptr(), 256, B, 4, 64, L, L, st))
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(); sys.exit(0)
    runs = [("auto", {})] + [(f"T{t}", {"AVSEP_GEMM_TILE": t, "AVSEP_LN_TILE": t[:5]}) for t in ("32x32x64", "32x64x64", "64x32x64", "64x64x64", "64x64x32")]
    res = {}
    for name, env in runs:
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        res[name] = json.loads(line[-1]) if line else {}
        if not line: print(name, "FAILED", r.stderr[-400:])
    keys = list(res["auto"].keys())
    print(f"{'case':>22s} " + " ".join(f"{n:>10s}" for n, _ in runs))
    for k in keys:
        print(f"{k:>22s} " + " ".join(f"{res[n].get(k) or 0:10.2f}" for n, _ in runs))
