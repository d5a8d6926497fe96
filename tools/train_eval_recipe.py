#!/usr/bin/env python3
"""The reference's end-to-end recipe on the drop-in (its caller, /root/reference/demo.py:121-200, is out of scope as
code; SURVEY.md §8(f) N3 asks that its metrics be reproducible): 500-item SyntheticAVDataset (1 s @ 8 kHz, 2 speakers),
d_model 128 / 4 heads / 2+2 layers / dropout 0.1, SNR of the untrained model on the first 20 items, 100 Adam steps
(batch 8 shuffled, lr 3e-4, SeparationLoss(0.5), clip 1.0), SNR again.  The reference prints input SNR ~3.2 dB and
a trained output SNR of 36-37 dB (README.md; 2.76 / 37.10 measured in SURVEY.md Appendix A.8); runs are unseeded there,
seeded here so the line can be compared across rounds."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402
from av_separation import AVSeparationTransformer, SyntheticAVDataset  # noqa: E402
from av_separation.evaluate import evaluate_separation  # noqa: E402
from av_separation.losses import SeparationLoss  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
ds = SyntheticAVDataset(num_samples=500, sample_rate=8000, duration=1.0, n_fft=512, hop_length=128, num_frames=25,
                        frame_h=32, frame_w=32, speaker_freqs=(220.0, 440.0))
item = ds[0]
F_, T_ = item["mixed_spec"].shape
model = AVSeparationTransformer(freq_bins=F_, d_model=128, nhead=4, num_encoder_layers=2, num_fusion_layers=2,
                                num_speakers=2, dropout=0.1).to(dev)
print(f"freq_bins={F_} T={T_} frames={item['lip_frames'].shape[0]} parameters={sum(p.numel() for p in model.parameters()):,}")
in0, out0 = evaluate_separation(model, ds, dev)
print(f"untrained: input SNR {in0:.2f} dB, output SNR {out0:.2f} dB")
loader = DataLoader(ds, batch_size=8, shuffle=True)
opt = torch.optim.Adam(model.parameters(), lr=3e-4)
crit = SeparationLoss(l1_weight=0.5)
model.train()
losses, t0 = [], time.perf_counter()
for step, batch in enumerate(loader, 1):
    opt.zero_grad()
    sep, _ = model(batch["mixed_spec"].to(dev), batch["lip_frames"].to(dev))
    loss = crit(sep, batch["clean_specs"].to(dev))
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    losses.append(loss.item())
    if step % 20 == 0:
        print(f"  step {step:4d} | loss {np.mean(losses[-20:]):.4f}")
    if step >= 100:
        break
torch.cuda.synchronize()
print(f"{len(losses)} training steps (one pass of the 500-item loader, as demo.py:96-113 does) in "
      f"{time.perf_counter() - t0:.2f} s, host data generation included")
in1, out1 = evaluate_separation(model, ds, dev)
print(f"trained:   input SNR {in1:.2f} dB, output SNR {out1:.2f} dB, improvement {out1 - in1:+.2f} dB "
      f"(reference README: 3.20 -> 37.24)")
