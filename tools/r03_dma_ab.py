#!/usr/bin/env python3
"""Developer tool (GPU box): LDS-DMA staging (gemm_dma_kernel, AVSEP_GEMM_DMA=1) against the register-ring GEMM on the
large shapes of configs 3-5 and the training step, one subprocess per variant, interleaved rounds."""
import os
os.environ.setdefault("AVSEP_LIB", "dev")
import ctypes as C, json, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
SHAPES = [(16064, 2048, 512), (16064, 512, 512), (16064, 1536, 512), (16064, 512, 2048), (16032, 4096, 512),
          (4016, 2048, 512), (4016, 512, 2048), (65536, 4096, 2048)]


def child():
    import torch
    from av_separation import _native
    lib = _native.load(); dev = torch.device("cuda:0"); out = {}
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (M, N, K) in SHAPES:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05
        b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
        for _ in range(3):
            rc = lib.avsep_op_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), M, N, K, 0, st)
        if rc != 0:
            out[f"{M}x{N}x{K}"] = None; continue
        n = 20 if M * N * K < 1e11 else 5
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                lib.avsep_op_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), M, N, K, 0, st)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / n * 1e3)
        out[f"{M}x{N}x{K}"] = best
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(); sys.exit(0)
    variants = [("ring 128x64x32", {"AVSEP_GEMM_TILE": "128x64x32"}), ("dma 128x64x32", {"AVSEP_GEMM_TILE": "128x64x32", "AVSEP_GEMM_DMA": "1"}),
                ("ring 128x64x16", {"AVSEP_GEMM_TILE": "128x64x16"}), ("dma 128x64x16", {"AVSEP_GEMM_TILE": "128x64x16", "AVSEP_GEMM_DMA": "1"}),
                ("ring 64x64x32", {"AVSEP_GEMM_TILE": "64x64x32"}), ("dma 64x64x32", {"AVSEP_GEMM_TILE": "64x64x32", "AVSEP_GEMM_DMA": "1"})]
    res = {n: {} for n, _ in variants}
    for rnd in range(2):
        for name, env in variants:
            r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env), capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            d = json.loads(line[-1]) if line else {}
            for k, v in d.items():
                if v is not None:
                    res[name][k] = min(res[name].get(k, 1e9), v)
    print(f"{'shape (us, TFLOP/s)':>20s} " + " ".join(f"{n:>22s}" for n, _ in variants))
    for (M, N, K) in SHAPES:
        k = f"{M}x{N}x{K}"
        print(f"{k:>20s} " + " ".join(f"{res[n].get(k, 0):12.1f} {2*M*N*K/max(res[n].get(k, 1e9), 1e-9)/1e6:8.1f}" for n, _ in variants))
