#!/usr/bin/env python3
"""Developer tool (GPU box): time avsep_op_linear for every instantiated tile on the GEMM shapes of a
workload.  Tiles are forced through the AVSEP_GEMM_TILE developer override, one subprocess per tile."""
import os
os.environ.setdefault("AVSEP_LIB", "dev")   # developer switches live in libavsep_hip_dev.so only
import ctypes as C, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
TILES = ["256x128x32", "128x128x32", "128x128x32/nopf", "128x64x32", "128x64x16", "64x64x32", "64x64x64", "64x32x32", "64x32x64", "32x32x32", "32x32x64"]
SHAPES = [(2016, 256, 256), (2016, 256, 1024), (2016, 768, 256), (2016, 1024, 256), (2016, 512, 256),
          (2016, 514, 512), (1600, 256, 256), (1600, 768, 256), (1600, 1024, 256), (1600, 256, 1024),
          (1600, 256, 128), (3200, 512, 512), (3200, 2048, 512), (3200, 512, 2048),
          (16064, 512, 512), (16064, 1536, 512), (16064, 2048, 512), (16064, 512, 2048), (16064, 514, 1024),
          (16032, 4096, 512), (4016, 512, 512), (4016, 1536, 512), (4016, 2048, 512), (4016, 512, 2048), (8032, 512, 512),
          (8032, 2048, 512), (8032, 512, 2048)]
if os.environ.get("SWEEP_BIG"):
    SHAPES = [s_ for s_ in SHAPES if s_[0] >= 3200]

def child():
    import torch
    from av_separation import _native
    lib = _native.load()
    dev = torch.device("cuda:0")
    out = {}
    for (M, N, K) in SHAPES:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05
        b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(5):
            rc = lib.avsep_op_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), M, N, K, 0, st)
        if rc != 0:
            out[f"{M}x{N}x{K}"] = None; continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            lib.avsep_op_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), M, N, K, 0, st)
        e1.record(); torch.cuda.synchronize()
        out[f"{M}x{N}x{K}"] = e0.elapsed_time(e1) / n * 1e3
    print(json.dumps(out))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(); sys.exit(0)
    if os.environ.get("SWEEP_VAR"):      # sweep the values of one developer environment switch instead of the tiles
        var, vals = os.environ["SWEEP_VAR"], os.environ["SWEEP_VALS"].split(",")
        res = {}
        for v in vals:
            env = dict(os.environ)
            if v != "unset": env[var] = v
            r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            res[v] = json.loads(line[-1]) if line else {}
        print(f"{var:>18s} " + " ".join(f"{v:>10s}" for v in vals))
        for (M, N, K) in SHAPES:
            k = f"{M}x{N}x{K}"
            print(f"{k:>18s} " + " ".join(f"{res[v].get(k) or 0:10.1f}" for v in vals) + "   TF: " +
                  " ".join(f"{2*M*N*K/(res[v].get(k) or 1e9)/1e6:6.1f}" for v in vals))
        sys.exit(0)
    res = {}
    for t in TILES + ["auto"]:
        env = dict(os.environ)
        if t != "auto":
            env["AVSEP_GEMM_TILE"] = t.split("/")[0]
            if t.endswith("/nopf"): env["AVSEP_G32_PF"] = "0"
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        res[t] = json.loads(line[-1]) if line else {}
        if not line: print(t, "FAILED", r.stderr[-500:])
    print(f"{'shape':>18s} " + " ".join(f"{t:>15s}" for t in TILES + ['auto']) + "   best  TF(best)")
    for (M, N, K) in SHAPES:
        k = f"{M}x{N}x{K}"
        vals = [res[t].get(k) for t in TILES + ["auto"]]
        best = min((v, t) for v, t in zip(vals[:-1], TILES) if v)
        print(f"{k:>18s} " + " ".join(f"{v:15.1f}" if v else f"{'-':>15s}" for v in vals) + f"   {best[1]:>15s} {2*M*N*K/best[0]/1e6:6.1f}")
