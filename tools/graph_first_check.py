#!/usr/bin/env python3
"""GPU check (tests/test_gpu_parity.py::test_graph_capture_as_the_first_forward_of_a_process): the first forward of a fresh
process is a hipGraph capture; the replay must equal the eager forward."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
import av_separation as av
dev = torch.device("cuda:0")
torch.manual_seed(0)
for d, T, N in ((512, 251, 50), (256, 63, 50)):
    m = av.AVSeparationTransformer(257, d, 8, 2, 2, 2, dropout=0.0).to(dev).eval()
    B = 16
    mx = torch.rand(B, 257, T, device=dev); lp = torch.rand(B, N, 32, 32, device=dev)
    mk = torch.empty(B, T, 2, 257, device=dev); sp = torch.empty(B, T, 2, 257, device=dev)
    m.run_static(mx, lp, mk, sp, graph=True)        # the FIRST forward of the process / model is a graph capture
    m.run_static(mx, lp, mk, sp, graph=True)
    torch.cuda.synchronize()
    with torch.no_grad():
        sep, masks = m(mx, lp)
    print(d, "graph-first == eager:", bool(torch.equal(mk.permute(0, 2, 3, 1), masks)), float(masks.min()), float(masks.max()))
