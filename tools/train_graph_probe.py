#!/usr/bin/env python3
"""Developer probe (GPU box): what would the cfg4 training step cost with NO host time?  The whole step (forward, PIT
loss, backward, clip, Adam) is captured into one graph -- with the dropout seeds frozen at capture time, so this is a
TIMING probe, not a training mode -- and replayed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "av-separation-transformer_amd")):
    sys.path.insert(0, p)
import torch
import av_separation as av
import bench
from av_separation import _train as tr
from av_separation.losses import SeparationLoss
dev = torch.device("cuda:0")
wl = bench.WORKLOADS["cfg4"]; B = wl["batch"]
torch.manual_seed(0)
model = av.AVSeparationTransformer(dropout=0.1, **wl["model"]).to(dev).train()
ds = av.SyntheticAVDataset(num_samples=B, **wl["data"])
items = [ds[i] for i in range(B)]
mixed = torch.stack([it["mixed_spec"] for it in items]).to(dev); lips = torch.stack([it["lip_frames"] for it in items]).to(dev)
targets = torch.stack([it["clean_specs"] for it in items]).to(dev)
crit = SeparationLoss(0.5)
opt = torch.optim.Adam(model.parameters(), lr=3e-4, fused=True, capturable=True)
_orig = tr.make_drop
tr.make_drop = lambda module, probs, seed=None, group=None: _orig(module, probs, seed=1234, group=group)   # no host draw
def step():
    opt.zero_grad(set_to_none=True)
    sep, _ = model(mixed, lips)
    loss = crit(sep, targets)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0, foreach=True)
    opt.step()
    return loss
def timed(f, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
print(f"eager: {timed(step, 10):.2f} ms/step")
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    loss = step()
print(f"graph replay: {timed(g.replay, 10):.2f} ms/step   loss {float(loss):.4f}")
for side in (True,):
    tr.SIDE_STREAM_WGRAD = side
    g2 = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    try:
        with torch.cuda.graph(g2):
            loss = step()
        print(f"graph replay, parameter gradients on a second stream: {timed(g2.replay, 10):.2f} ms/step")
    except Exception as e:
        print("side-stream capture failed:", repr(e)[:300])
