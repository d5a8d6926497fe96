#!/usr/bin/env python3
"""Developer diagnostic (GPU box, developer library): long-sequence attention against the number of resident workgroups per
CU (capped by extra dynamic LDS, AVSEP_ATTN_PAD_LDS) -- does the matrix-pipe share grow with the waves per SIMD?
Needs the launch of attention_lds_kernel<2, 2, false> in attention.hip to pass `dev_env("AVSEP_ATTN_PAD_LDS")` bytes as its
dynamic shared memory (a three-line developer patch, not kept in the sources; result in profiles/r03_mfma_valu_exclusive.txt)."""
import ctypes as C, os, sys
os.environ["AVSEP_LIB"] = "dev"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
from av_separation import _native
lib = _native.load(); dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
h, dh = 8, 64
for (B, L) in ((192, 501), (192, 251)):
    d = h * dh
    qkv = torch.randn(B * L, 3 * d, device=dev); o = torch.empty(B * L, d, device=dev)
    f = lambda: lib.avsep_op_attention(qkv.data_ptr(), 3 * d, qkv.data_ptr() + 4 * d, 3 * d, qkv.data_ptr() + 8 * d, 3 * d, o.data_ptr(), d, B, h, dh, L, L, st)
    for pad, wg in ((100000, 1), (40000, 2), (0, 3)):
        if pad: os.environ["AVSEP_ATTN_PAD_LDS"] = str(pad)
        else: os.environ.pop("AVSEP_ATTN_PAD_LDS", None)
        for _ in range(3): rc = f()
        assert rc == 0, lib.avsep_last_error()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"L={L} B={B}: {wg} workgroup(s) per CU = {wg} wave(s) per SIMD: {us:9.1f} us  {4.0*B*h*L*L*dh/us/1e6:6.1f} TFLOP/s", flush=True)
