#!/usr/bin/env python3
"""Bit-identity check of the chained schedules (developer build of the library: run with AVSEP_LIB=dev; tests/test_gpu_parity.py
does).  For every (schedule, group, skew) and several batch sizes the fused forward's outputs must equal the launch-per-op
schedule's bit for bit, eagerly and under graph replay, and no dependency wait may have timed out."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "av-separation-transformer_amd")):
    sys.path.insert(0, p)
import torch
import av_separation as av
dev = torch.device("cuda:0")
torch.manual_seed(3)
m = av.AVSeparationTransformer(dropout=0.0).to(dev).eval()
cases = [(1, 8, 0.0), (1, 4, 1.0), (1, 1, 2.5), (2, 8, 0.0), (2, 1, 0.5), (2, 2, 2.0)]
n = 0
for B in (32, 5, 11):
    ds = av.SyntheticAVDataset(num_samples=B)
    items = [ds[i] for i in range(B)]
    mixed = torch.stack([x["mixed_spec"] for x in items]).to(dev)
    lips = torch.stack([x["lip_frames"] for x in items]).to(dev)
    with torch.no_grad():
        m.set_schedule(0)
        sep0, masks0 = m(mixed, lips)
        for schedule, group, skew in cases:
            m.set_schedule(schedule, group, skew)
            for _ in range(3):                       # counters are re-zeroed by every launch
                sep1, masks1 = m(mixed, lips)
                m.chain_status()
                assert torch.equal(masks1, masks0) and torch.equal(sep1, sep0), (B, schedule, group, skew)
            m.enable_graph_replay(True)
            for _ in range(3):
                sg, mg = m(mixed, lips)
                m.chain_status()
                assert torch.equal(mg, masks0) and torch.equal(sg, sep0), (B, schedule, group, skew, "graph")
            m.enable_graph_replay(False)
            n += 1
        m.set_schedule(0)
print(f"CHAIN_CHECK_OK {n} cases")
