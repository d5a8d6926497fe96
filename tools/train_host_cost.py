#!/usr/bin/env python3
"""Is the eager training step host-bound?  Enqueue time of K steps (no sync inside) vs wall time including the final sync."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "av-separation-transformer_amd")):
    sys.path.insert(0, p)
import torch
import av_separation as av
import bench
from av_separation.losses import SeparationLoss
dev = torch.device("cuda:0")
wl = bench.WORKLOADS["cfg4"]; B = wl["batch"]
torch.manual_seed(0)
model = av.AVSeparationTransformer(dropout=0.1, **wl["model"]).to(dev).train()
ds = av.SyntheticAVDataset(num_samples=B, **wl["data"])
items = [ds[i] for i in range(B)]
mixed = torch.stack([it["mixed_spec"] for it in items]).to(dev); lips = torch.stack([it["lip_frames"] for it in items]).to(dev)
targets = torch.stack([it["clean_specs"] for it in items]).to(dev)
crit = SeparationLoss(0.5); opt = torch.optim.Adam(model.parameters(), lr=3e-4, fused=True)
def step(loss_sync=True):
    opt.zero_grad(set_to_none=False)
    sep, _ = model(mixed, lips)
    loss = crit(sep, targets) if loss_sync else (sep - targets).abs().mean()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0, foreach=True)
    opt.step()
for name, ls in (("PIT loss (host compares the permutation losses: one sync per step)", True), ("L1-only loss (no sync)", False)):
    for _ in range(3): step(ls)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): step(ls)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: enqueue {1e3*(t1-t0)/10:.2f} ms/step, wall {1e3*(t2-t0)/10:.2f} ms/step")
