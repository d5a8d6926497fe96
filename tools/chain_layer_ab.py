#!/usr/bin/env python3
"""Developer tool (GPU box): the audio encoder ALONE on the chip (AudioEncoder.forward, model.py:54-60 -- transpose, two Conv1d GEMMs,
then the two pre-norm layers) under the three launch schedules: 0 = ten launches for the layers, 1 / 2 = ONE dependency-driven
persistent launch (one queue + write-through hand-offs / XCD-local queues + plain stores).  Same run, alternating, bit-identity
checked.  With AVSEP_LIB=dev AVSEP_CHAIN_DBG=1 the library prints the per-op phase stamps of the chained launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "av-separation-transformer_amd")):
    sys.path.insert(0, p)
import torch
import av_separation as av
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
torch.manual_seed(5)
enc = av.AudioEncoder(257, 256, 4, 2, dropout=0.0).to(dev).eval()
ds = av.SyntheticAVDataset(num_samples=B)
mixed = torch.stack([ds[i]["mixed_spec"] for i in range(B)]).to(dev)
cases = [(0, 8, 0.0), (1, 8, 0.0), (2, 8, 0.0), (2, 1, 0.5), (2, 2, 2.5)]
ref = None
with torch.no_grad():
    for rnd in range(3):
        for sched, group, skew in cases:
            enc._engine.set_schedule(sched, group, skew)
            out = enc(mixed)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            same = bool(torch.equal(out, ref))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                enc(mixed)
            e1.record()
            torch.cuda.synchronize()
            print(f"round {rnd}  schedule {sched} group {group} skew {skew:g}: {e0.elapsed_time(e1) / reps * 1e3:8.1f} us per encoder forward, "
                  f"bit-identical to schedule 0: {same}", flush=True)
