#!/usr/bin/env python3
"""Experiment (not the headline): how much faster does the cfg2 clip stream go when R independent batches are in
flight at once (R model replicas, each replaying its own hipGraph on its own stream)?  One step per replica per
iteration; reports ms per 32-clip step.  Shows how far a single batch of 32 is from saturating the chip."""
import copy
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
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
B = wl["batch"]
torch.manual_seed(0)
base = av.AVSeparationTransformer(dropout=0.0, **wl["model"]).to(dev).eval()
ds = av.SyntheticAVDataset(num_samples=B, **wl["data"])
items = [ds[i] for i in range(B)]
mixed = torch.stack([it["mixed_spec"] for it in items]).to(dev).contiguous()
lips = torch.stack([it["lip_frames"] for it in items]).to(dev).contiguous()
_, F, T = mixed.shape
S = wl["model"]["num_speakers"]
GRAPH = os.environ.get("INFLIGHT_GRAPH", "1") == "1"
for R in (1, 2, 3):
    SHARED = os.environ.get("INFLIGHT_SHARED") == "1"          # one context, R workspaces (slots) instead of R replicas
    models = [base] * R if SHARED else [base] + [copy.deepcopy(base) for _ in range(R - 1)]
    ins = [(mixed.clone(), lips.clone()) for _ in range(R)]
    outs = [(torch.empty(B, T, S, F, device=dev), torch.empty(B, T, S, F, device=dev)) for _ in range(R)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(R)]
    with torch.no_grad():
        for _ in range(10):
            for i, (m, (mk, sp), st) in enumerate(zip(models, outs, streams)):
                with torch.cuda.stream(st):
                    m.run_static(ins[i][0], ins[i][1], mk, sp, graph=GRAPH, slot=i if SHARED else 0)
        torch.cuda.synchronize()
        steps = 100
        t0 = time.perf_counter()
        for _ in range(steps):
            for i, (m, (mk, sp), st) in enumerate(zip(models, outs, streams)):
                with torch.cuda.stream(st):
                    m.run_static(ins[i][0], ins[i][1], mk, sp, graph=GRAPH, slot=i if SHARED else 0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    same = all(torch.equal(outs[0][0], o[0]) for o in outs[1:])
    print(f"[{'graph' if GRAPH else 'eager'}{' shared ctx' if SHARED else ''}] replicas in flight {R}: {dt / (steps * R) * 1e3:.4f} ms per {B}-clip step, {B * steps * R / dt:9.0f} clips/s, "
          f"outputs identical across replicas: {same}", flush=True)
