#!/usr/bin/env python3
"""Developer tool (GPU box): the pre-split GEMM (csrc/gemm_planes.hip: operands as bf16 planes, LDS-DMA staging) against the
split-precision kernels that cut their operands in flight (gemm_split.hip), on the model's large shapes: bit equality of the
fp32 output and of the plane output, time (interleaved rounds in one process), error against float64."""
import os
os.environ.setdefault("AVSEP_LIB", "dev")
import ctypes as C, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
from av_separation import _native
lib = _native.load(); dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ROUNDS = int(os.environ.get("PROBE_ROUNDS", "5"))
VARIANTS = [int(v) for v in os.environ.get("PROBE_VARIANTS", "0").split(",")]   # AVSEP_PLANES_V of the developer library
N_IT = int(os.environ.get("PROBE_ITERS", "20"))


def planes_of(x, rows=None):
    M, K = x.shape
    rows = rows or M
    P = torch.zeros(K // 32 * 3 * rows * 32, dtype=torch.int16, device=dev)
    assert lib.avsep_op_split_planes(x.data_ptr(), x.stride(0), P.data_ptr(), rows, M, K, st) == 0, lib.avsep_last_error()
    return P


def time_rounds(fns):
    """interleaved rounds: every variant N_IT times per round; median per variant"""
    for f in fns:
        for _ in range(4): f()
    torch.cuda.synchronize()
    ts = [[] for _ in fns]
    for _ in range(ROUNDS):
        for i, f in enumerate(fns):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(N_IT): f()
            torch.cuda.synchronize(); ts[i].append((time.perf_counter() - t0) / N_IT)
    return [sorted(t)[len(t) // 2] for t in ts], [min(t) for t in ts]


torch.manual_seed(0)
shapes = ((16064, 2048, 512, 1, False), (16064, 1536, 512, 0, False), (16064, 512, 512, 0, True), (16064, 512, 2048, 0, True),
          (16032, 2048, 512, 2, False), (8192, 1024, 1024, 0, False), (4016, 2048, 512, 0, False), (4016, 512, 2048, 0, True),
          (3200, 2048, 512, 1, False), (3200, 512, 2048, 0, True), (1004, 2048, 512, 1, False), (251, 1536, 512, 0, False),
          (777, 640, 96, 2, False), (300, 128, 32, 0, False), (515, 256, 64, 1, True))
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes = shapes[:4]
print("shape (M, N, K) act res | split (in-flight): us TFLOP/s | planes: us TFLOP/s (min) | planes -> planes: us | speed-up | bits | err/max|y|")
for M, N, K, act, res in shapes:
    x = (torch.randn(M, K, device=dev) * 2 + 0.7); w = torch.randn(N, K, device=dev) * 0.06; b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    xp, wp = planes_of(x, M + 5), planes_of(w)
    y0 = torch.empty(M, N, device=dev); y1 = torch.full((M, N), float("nan"), device=dev)
    rp = r.data_ptr() if res else None
    f_split = lambda: lib.avsep_op_linear_split(x.data_ptr(), w.data_ptr(), b.data_ptr(), rp, y0.data_ptr(), M, N, K, act, st)
    def f_planes(v=None):
        if v is not None: os.environ["AVSEP_PLANES_V"] = str(v)
        return lib.avsep_op_linear_planes(xp.data_ptr(), M + 5, wp.data_ptr(), N, b.data_ptr(), rp, y1.data_ptr(), None, 0, M, N, K, act, st)
    assert f_split() == 0, lib.avsep_last_error()
    bits = True
    for v in VARIANTS:
        y1.fill_(float("nan"))
        assert f_planes(v) == 0, lib.avsep_last_error()
        torch.cuda.synchronize()
        if v not in (10, 12, 34, 40): bits = bits and torch.equal(y0, y1)      # 10 / 12: timing ablations (no DMA / no MFMA in the loop)
    fns = [f_split] + [(lambda v=v: f_planes(v)) for v in VARIANTS[1:]] + [lambda: f_planes(VARIANTS[0])]
    os.environ["AVSEP_PLANES_V"] = str(VARIANTS[0])
    bits_p = "-"
    if not res and N % 32 == 0:
        yp = torch.zeros(N // 32 * 3 * M * 32, dtype=torch.int16, device=dev)
        f_pp = lambda: lib.avsep_op_linear_planes(xp.data_ptr(), M + 5, wp.data_ptr(), N, b.data_ptr(), None, None, yp.data_ptr(), M, M, N, K, act, st)
        assert f_pp() == 0, lib.avsep_last_error()
        torch.cuda.synchronize()
        bits_p = torch.equal(yp, planes_of(y0))
        fns.append(f_pp)
    ref = x.double() @ w.double().t() + b.double()
    ref = {0: ref, 1: torch.relu(ref), 2: torch.nn.functional.gelu(ref)}[act]
    if res: ref = ref + r.double()
    err = float((y1.double() - ref).abs().max()) / float(ref.abs().max())
    del ref
    nv = len(VARIANTS)
    med, mn = time_rounds(fns)
    fl = 2.0 * M * N * K
    pp = f"{med[nv + 1] * 1e6:8.1f}" if len(med) > nv + 1 else "       -"
    others = " ".join(f"v{v}:{fl / med[1 + i] / 1e12:6.1f}" for i, v in enumerate(VARIANTS[1:]))
    print(f"({M:6d},{N:5d},{K:5d}) {act} {int(res)} | {med[0] * 1e6:8.1f} {fl / med[0] / 1e12:6.1f} | {med[nv] * 1e6:8.1f} {fl / med[nv] / 1e12:6.1f} ({fl / mn[nv] / 1e12:6.1f}) | {pp} | x{med[0] / med[nv]:.3f} | y {bits} planes {bits_p} | {err:.2e} | {others}", flush=True)
