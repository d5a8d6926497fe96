#!/usr/bin/env python3
"""Developer experiment: how does the replayed cfg2 step respond to removing layers?  (Is the step time the critical
path of the kernels, or something else?)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "av-separation-transformer_amd")):
    sys.path.insert(0, p)
import torch
import av_separation as av
dev = torch.device("cuda:0")
B = 32
mixed = torch.rand(B, 257, 63, device=dev); lips = torch.rand(B, 50, 32, 32, device=dev)
for Le, Lf in ((2, 2), (2, 1), (2, 0), (1, 2), (0, 2), (0, 0)):
    torch.manual_seed(0)
    m = av.AVSeparationTransformer(num_encoder_layers=Le, num_fusion_layers=Lf, dropout=0.0).to(dev).eval()
    mk, sp = torch.empty(B, 63, 2, 257, device=dev), torch.empty(B, 63, 2, 257, device=dev)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st), torch.no_grad():
        for _ in range(20):
            m.run_static(mixed, lips, mk, sp, graph=True)
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            m.run_static(mixed, lips, mk, sp, graph=True)
        t1 = time.perf_counter()
        st.synchronize()
        dt = (time.perf_counter() - t0) / 200
        # two graph execs (two output buffer sets) alternating: does re-launching ONE exec serialise on the host?
        mk2, sp2 = torch.empty_like(mk), torch.empty_like(sp)
        for _ in range(4):
            m.run_static(mixed, lips, mk, sp, graph=True); m.run_static(mixed, lips, mk2, sp2, graph=True)
        st.synchronize()
        t2 = time.perf_counter()
        for _ in range(100):
            m.run_static(mixed, lips, mk, sp, graph=True); m.run_static(mixed, lips, mk2, sp2, graph=True)
        st.synchronize()
        dt2 = (time.perf_counter() - t2) / 200
    print(f"encoder layers {Le}, fusion layers {Lf}: {dt * 1e6:7.1f} us/step (host enqueue {(t1 - t0) / 200 * 1e6:6.1f}), "
          f"alternating two execs {dt2 * 1e6:7.1f} us/step", flush=True)
