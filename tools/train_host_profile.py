#!/usr/bin/env python3
"""Developer tool (GPU box): where the HOST time of one cfg4 training step goes.  Prints (1) the wall time of a step when the host
never waits for the GPU inside it (enqueue time) next to the synchronised step time, (2) a cProfile of 5 steps by cumulative time."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
import av_separation as av
from av_separation.losses import SeparationLoss
import bench
dev = torch.device("cuda:0")
wl = bench.WORKLOADS["cfg4"]; mk, dk = wl["model"], wl["data"]; B = wl["batch"]
torch.manual_seed(0)
model = av.AVSeparationTransformer(dropout=0.1, **mk).to(dev).train()
ds = av.SyntheticAVDataset(num_samples=B, sample_rate=dk["sample_rate"], duration=dk["duration"], num_frames=dk["num_frames"],
                           frame_h=dk["frame_h"], frame_w=dk["frame_w"], speaker_freqs=dk["speaker_freqs"])
its = [ds[i] for i in range(B)]
mixed = torch.stack([it["mixed_spec"] for it in its]).to(dev).contiguous()
lips = torch.stack([it["lip_frames"] for it in its]).to(dev).contiguous()
targets = torch.stack([it["clean_specs"] for it in its]).to(dev).contiguous()
crit = SeparationLoss(0.5)
opt = torch.optim.Adam(model.parameters(), lr=3e-4, fused=True)
def step():
    opt.zero_grad()
    sep, _ = model(mixed, lips)
    loss = crit(sep, targets)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0, foreach=True)
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
for label, n in (("10 steps", 10),):
    t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{label}: host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, with the final sync {1e3 * (t2 - t0) / n:.2f} ms/step")
# phases of the host time, each phase synchronised (so GPU time is excluded from the next one)
ph = {"zero_grad": 0.0, "forward": 0.0, "loss": 0.0, "backward": 0.0, "clip": 0.0, "adam": 0.0}
for _ in range(5):
    torch.cuda.synchronize(); t = time.perf_counter(); opt.zero_grad(); ph["zero_grad"] += time.perf_counter() - t
    torch.cuda.synchronize(); t = time.perf_counter(); sep, _ = model(mixed, lips); ph["forward"] += time.perf_counter() - t
    torch.cuda.synchronize(); t = time.perf_counter(); loss = crit(sep, targets); ph["loss"] += time.perf_counter() - t
    torch.cuda.synchronize(); t = time.perf_counter(); loss.backward(); ph["backward"] += time.perf_counter() - t
    torch.cuda.synchronize(); t = time.perf_counter(); torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0, foreach=True); ph["clip"] += time.perf_counter() - t
    torch.cuda.synchronize(); t = time.perf_counter(); opt.step(); ph["adam"] += time.perf_counter() - t
print("host time per phase (ms, enqueue only):", {k: round(1e3 * v / 5, 2) for k, v in ph.items()})
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
