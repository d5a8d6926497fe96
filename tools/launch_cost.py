#!/usr/bin/env python3
"""Is the cfg2 step host-bound?  Times (a) the host cost of enqueueing one graph replay (no sync inside the loop),
(b) the wall time per step with the queue kept full, (c) eager launches, for comparison."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "av-separation-transformer_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import av_separation as av  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
wl = bench.WORKLOADS["cfg2"]
B = wl["batch"]
torch.manual_seed(0)
m = av.AVSeparationTransformer(dropout=0.0, **wl["model"]).to(dev).eval()
mixed = torch.rand(B, 257, 63, device=dev)
lips = torch.rand(B, 50, 32, 32, device=dev)
mk, sp = torch.empty(B, 63, 2, 257, device=dev), torch.empty(B, 63, 2, 257, device=dev)
st = torch.cuda.Stream(device=dev)
for graph in (True, False):
    with torch.cuda.stream(st), torch.no_grad():
        for _ in range(20):
            m.run_static(mixed, lips, mk, sp, graph=graph)
        st.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            m.run_static(mixed, lips, mk, sp, graph=graph)
        t1 = time.perf_counter()
        st.synchronize()
        t2 = time.perf_counter()
        # host cost alone: enqueue a few, wait until the GPU is idle, repeat
        host = []
        for _ in range(50):
            st.synchronize()
            a = time.perf_counter()
            m.run_static(mixed, lips, mk, sp, graph=graph)
            host.append(time.perf_counter() - a)
        st.synchronize()
    host.sort()
    print(f"graph={graph}: enqueue loop {1e3 * (t1 - t0) / n:.4f} ms/step, wall {1e3 * (t2 - t0) / n:.4f} ms/step, "
          f"host cost of one enqueue on an idle queue: median {1e3 * host[len(host) // 2]:.4f} ms", flush=True)
