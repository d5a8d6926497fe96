#!/usr/bin/env python3
"""Developer tool (GPU box): every instance of the LayerNorm-fused GEMM on the cfg2 shapes (us per launch, back-to-back
launches on one stream; one subprocess per instance -- AVSEP_LN_TILE is read per call but keeps the run simple)."""
import os
os.environ.setdefault("AVSEP_LIB", "dev")
import ctypes as C, json, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
SHAPES = [(2016, 768), (2016, 1024), (1600, 768), (1600, 1024), (1008, 1024), (1008, 256), (2016, 256), (2016, 512), (1008, 512)]
TILES = ["32x32", "32x64", "64x32", "64x64", "128x64x16", "64x128x16", "64x96x8", "64x64x8", "128x32x8", "32x128x8"]


def child():
    import torch
    from av_separation import _native
    lib = _native.load(); dev = torch.device("cuda:0"); out = {}
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    K = 256
    for (M, N) in SHAPES:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
        g = torch.rand(K, device=dev) + 0.5; be = torch.randn(K, device=dev); y = torch.empty(M, N, device=dev)
        call = lambda: lib.avsep_op_ln_linear(x.data_ptr(), g.data_ptr(), be.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, M, N, K, 1, 1e-5, 1, st)
        if call() != 0:
            out[f"{M}x{N}"] = None; continue
        for _ in range(5): call()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): call()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
        out[f"{M}x{N}"] = best
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(); sys.exit(0)
    res = {}
    for t in TILES + ["auto"]:
        env = dict(os.environ)
        if t != "auto": env["AVSEP_LN_TILE"] = t
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        res[t] = json.loads(line[-1]) if line else {}
        if not line: print(t, "FAILED", r.stderr[-300:])
    print(f"{'MxN (K=256), us':>16s} " + " ".join(f"{t:>10s}" for t in TILES + ['auto']))
    for (M, N) in SHAPES:
        k = f"{M}x{N}"
        print(f"{k:>16s} " + " ".join(f"{(res[t].get(k) or 0):10.2f}" for t in TILES + ['auto']))
