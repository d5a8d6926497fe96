#!/usr/bin/env python3
"""Developer tool (GPU box): the two-term fp16 GEMM (csrc/gemm_h2.hip) against the three-term bf16 kernels (in-flight split and
pre-split planes) on the model's large shapes: time (interleaved rounds in one process), error against float64, bit equality of
the developer variants (AVSEP_H2_KERNEL2, AVSEP_H2_TILE=64)."""
import os
os.environ.setdefault("AVSEP_LIB", "dev")
import ctypes as C, sys, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
from av_separation import _native
lib = _native.load(); dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ROUNDS = int(os.environ.get("PROBE_ROUNDS", "5")); N_IT = int(os.environ.get("PROBE_ITERS", "20"))


def h2_exp(bound): return 14 - math.frexp(bound * (1 + 1e-5))[1] if bound > 0 else 0


def time_rounds(fns):
    for f in fns:
        for _ in range(4): f()
    torch.cuda.synchronize()
    ts = [[] for _ in fns]
    for _ in range(ROUNDS):
        for i, f in enumerate(fns):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(N_IT): f()
            torch.cuda.synchronize(); ts[i].append((time.perf_counter() - t0) / N_IT)
    return [sorted(t)[len(t) // 2] for t in ts]


torch.manual_seed(0)
shapes = ((16064, 2048, 512, 1, False), (16064, 1536, 512, 0, False), (16064, 512, 512, 0, True), (16064, 512, 2048, 0, True),
          (8032, 2048, 512, 2, False), (8032, 512, 2048, 0, True), (3200, 2048, 512, 1, False), (3200, 512, 2048, 0, True),
          (1600, 2048, 512, 1, False), (1600, 512, 2048, 0, True), (251, 1536, 512, 0, False), (251, 512, 2048, 0, True))
print("shape (M, N, K) act res | bf16x3 in flight: us TF | fp16x2: us TF err | 128x128 2WG/CU: us TF bits | 64x64: us TF bits | fp32 err")
for M, N, K, act, res in shapes:
    x = (torch.randn(M, K, device=dev) * 2 + 0.7); w = torch.randn(N, K, device=dev) * 0.06; b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    rp = r.data_ptr() if res else None
    ex = h2_exp(8.0 * float(x.abs().max()))
    xp = torch.zeros(K // 32 * 2 * M * 32, dtype=torch.int16, device=dev)
    assert lib.avsep_op_split_h2(x.data_ptr(), K, xp.data_ptr(), M, M, K, None, ex, st) == 0
    ew = torch.zeros(N, dtype=torch.int32, device=dev); l2 = torch.zeros(N, device=dev)
    assert lib.avsep_op_h2_row_stats(w.data_ptr(), N, K, ew.data_ptr(), l2.data_ptr(), st) == 0
    wp = torch.zeros(K // 32 * 2 * N * 32, dtype=torch.int16, device=dev)
    assert lib.avsep_op_split_h2(w.data_ptr(), K, wp.data_ptr(), N, N, K, ew.data_ptr(), 0, st) == 0
    cs = torch.ldexp(torch.ones(N, device=dev), -(ew + ex))
    y0 = torch.empty(M, N, device=dev); ys = [torch.full((M, N), float("nan"), device=dev) for _ in range(3)]
    f_split = lambda: lib.avsep_op_linear_split(x.data_ptr(), w.data_ptr(), b.data_ptr(), rp, y0.data_ptr(), M, N, K, act, st)
    def f_h2(i, env):
        for k in ("AVSEP_H2_KERNEL2", "AVSEP_H2_TILE", "AVSEP_H2_MID"): os.environ.pop(k, None)
        os.environ.update(env)
        return lib.avsep_op_linear_h2(xp.data_ptr(), M, wp.data_ptr(), N, cs.data_ptr(), None, b.data_ptr(), rp, ys[i].data_ptr(), None, 0, 0, M, N, K, act, st)
    envs = ({}, {"AVSEP_H2_MID": "1"}, {"AVSEP_H2_TILE": "64"})
    assert f_split() == 0
    for i, e in enumerate(envs): assert f_h2(i, e) == 0, lib.avsep_last_error()
    torch.cuda.synchronize()
    ref = x.double() @ w.double().t() + b.double()
    ref = {0: ref, 1: torch.relu(ref), 2: torch.nn.functional.gelu(ref)}[act]
    if res: ref = ref + r.double()
    sc = float(ref.abs().max())
    e32 = float((y0.double() - ref).abs().max()) / sc; eh = float((ys[0].double() - ref).abs().max()) / sc
    bits = [torch.equal(ys[0], ys[i]) for i in (1, 2)]
    del ref
    med = time_rounds([f_split] + [(lambda i=i, e=e: f_h2(i, e)) for i, e in enumerate(envs)])
    fl = 2.0 * M * N * K
    print(f"({M:6d},{N:5d},{K:5d}) {act} {int(res)} | {med[0]*1e6:7.1f} {fl/med[0]/1e12:6.1f} | {med[1]*1e6:7.1f} {fl/med[1]/1e12:6.1f} {eh:.2e} | "
          f"{med[2]*1e6:7.1f} {fl/med[2]/1e12:6.1f} {bits[0]} | {med[3]*1e6:7.1f} {fl/med[3]/1e12:6.1f} {bits[1]} | bf16x3 err {e32:.2e}", flush=True)
