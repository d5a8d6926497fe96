#!/usr/bin/env python3
"""Developer tool (GPU box): time avsep_op_wgrad_bias_direct (weight + bias gradient of a Linear) on the training
shapes for several (tile, slice count) choices forced through AVSEP_WGRAD_TILE / AVSEP_WGRAD_SLICES, one subprocess
per choice; 'auto' = the library's own choice.  Prints us per call (launch + slice sum) and TFLOP/s."""
import os
os.environ.setdefault("AVSEP_LIB", "dev")   # developer switches live in libavsep_hip_dev.so only
import ctypes as C, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
SHAPES = [(512, 512, 4016), (1536, 512, 4016), (2048, 512, 4016), (512, 2048, 4016), (1024, 512, 4016),
          (512, 512, 1200), (2048, 512, 1200), (512, 2048, 1200), (1536, 512, 1200), (512, 128, 1200)]
CHOICES = ["auto", "32/1", "32/2", "32/4", "32/8", "64/1", "64/2", "64/4", "64/8", "64/16"]

def child():
    import torch
    from av_separation import _native
    lib = _native.load(); dev = torch.device("cuda:0"); out = {}
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (N, K, R) in SHAPES:
        dy = torch.randn(R, N, device=dev); x = torch.randn(R, K, device=dev); buf = torch.empty(N * K + N, device=dev)
        ns = lib.avsep_op_wgrad_bias_direct_scratch_floats(N, K, R)
        scr = torch.empty(max(ns, 1), device=dev)
        f = lambda: lib.avsep_op_wgrad_bias_direct(dy.data_ptr(), N, x.data_ptr(), K, buf.data_ptr(), scr.data_ptr(), N, K, R, st)
        for _ in range(5): rc = f()
        if rc != 0: out[f"{N}x{K}x{R}"] = None; continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        out[f"{N}x{K}x{R}"] = e0.elapsed_time(e1) / 50 * 1e3
    print(json.dumps(out))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(); sys.exit(0)
    res = {}
    for c in CHOICES:
        env = dict(os.environ)
        if c != "auto":
            env["AVSEP_WGRAD_TILE"], env["AVSEP_WGRAD_SLICES"] = c.split("/")
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        res[c] = json.loads(line[-1]) if line else {}
    print(f"{'N x K x R':>18s} " + " ".join(f"{c:>8s}" for c in CHOICES) + "   best  TF(best) TF(auto)")
    for (N, K, R) in SHAPES:
        k = f"{N}x{K}x{R}"
        vals = [res[c].get(k) for c in CHOICES]
        best = min((v, c) for v, c in zip(vals[1:], CHOICES[1:]) if v)
        print(f"{k:>18s} " + " ".join(f"{v:8.1f}" if v else f"{'-':>8s}" for v in vals) + f"   {best[1]:>6s} {2.0*N*K*R/best[0]/1e6:6.1f} {2.0*N*K*R/(vals[0] or 1e9)/1e6:6.1f}")
