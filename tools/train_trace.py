#!/usr/bin/env python3
"""Developer tool (GPU box): which host operation is behind every GPU kernel of one cfg4 training step?  torch.profiler
over ONE step; prints launches per step grouped by (kernel, the innermost python frame of this repo that issued it)."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "av-separation-transformer_amd")):
    sys.path.insert(0, p)
import torch
import av_separation as av
import bench
from av_separation.losses import SeparationLoss
from torch.profiler import profile, ProfilerActivity

wl = bench.WORKLOADS["cfg4"]; B = int(os.environ.get("TRACE_BATCH", wl["batch"])); dev = torch.device("cuda:0")
torch.manual_seed(0)
m = av.AVSeparationTransformer(dropout=0.1, **wl["model"]).to(dev).train()
ds = av.SyntheticAVDataset(num_samples=B, **wl["data"]); it = [ds[i] for i in range(B)]
mixed = torch.stack([x["mixed_spec"] for x in it]).to(dev); lips = torch.stack([x["lip_frames"] for x in it]).to(dev)
tg = torch.stack([x["clean_specs"] for x in it]).to(dev)
crit = SeparationLoss(0.5); opt = torch.optim.Adam(m.parameters(), lr=3e-4, fused=True)


def step():
    opt.zero_grad()
    sep, _ = m(mixed, lips)
    loss = crit(sep, tg)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0, foreach=True)
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
by_corr = {}
for e in ev:
    if e.device_type.name == "CPU" and getattr(e, "kernels", None):
        for k in e.kernels:
            by_corr[id(k)] = e
cnt = collections.Counter(); dur = collections.Counter()
for e in ev:
    if e.device_type.name == "CPU" and e.kernels:
        frame = next((f for f in (e.stack or []) if "av_separation" in f or "bench.py" in f or "losses.py" in f), "(torch internals)")
        for k in e.kernels:
            key = (k.name.split("(")[0][-60:], e.name[:40], frame.split("/")[-1][:60])
            cnt[key] += 1; dur[key] += k.duration
print(f"{'launches':>8s} {'us':>9s}  kernel | host op | frame")
for key, n in sorted(cnt.items(), key=lambda kv: -dur[kv[0]])[:70]:
    print(f"{n:8d} {dur[key]:9.1f}  {key[0]} | {key[1]} | {key[2]}")
print("total launches", sum(cnt.values()))
