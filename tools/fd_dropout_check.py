#!/usr/bin/env python3
"""Developer diagnostic (GPU box): the fixed-mask finite-difference check of tests/test_train_gpu.py::
test_dropout_training_is_self_consistent for several mask seeds and step sizes (is a mismatch the mask or the difference?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd")); sys.path.insert(0, ROOT)
import torch
import av_separation as av
from av_separation._train import train_forward
from av_separation.losses import SeparationLoss
from oracle import seeded
dev = torch.device("cuda:0")
torch.manual_seed(3)
m = av.AVSeparationTransformer(freq_bins=33, d_model=64, nhead=4, num_encoder_layers=1, num_fusion_layers=1,
                               num_speakers=2, dropout=0.1).to(dev).train()
mx, lp = seeded.inputs(77, 2, 33, 12, 4, 8, 8)
mixed, lips = torch.from_numpy(mx).to(dev), torch.from_numpy(lp).to(dev)
tg = torch.rand(2, 2, 33, 12, device=dev) * mixed.unsqueeze(1)
crit0 = SeparationLoss(0.5)
wt = torch.randn(2, 2, 33, 12, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
crit = (lambda s, t: (s * wt).sum() / 64.0) if os.environ.get("FD_LINEAR") else crit0   # a linear functional of the outputs
params = [p for p in m.parameters()]
gen = torch.Generator(device="cpu").manual_seed(5)
dirs = [torch.randn(p.shape, generator=gen).to(dev) for p in params]
for seed in (11, 12, 13, 14, 15, 16):
    m.zero_grad()
    s1, _ = train_forward(m, mixed, lips, seed=seed)
    crit(s1, tg).backward()
    analytic = sum(float((p.grad * d_).sum()) for p, d_ in zip(params, dirs))
    out = []
    for eps in (4e-3, 2e-3, 1e-3, 5e-4, 2.5e-4):
        vals = []
        with torch.no_grad():
            for sgn in (+1, -1):
                for p, d_ in zip(params, dirs): p.add_(sgn * eps * d_)
                vals.append(float(crit(train_forward(m, mixed, lips, seed=seed)[0], tg).double()))
                for p, d_ in zip(params, dirs): p.sub_(sgn * eps * d_)
        out.append((vals[0] - vals[1]) / (2 * eps))
    print(f"seed {seed}: analytic {analytic:+.4f}; central differences at eps 4e-3 .. 2.5e-4: " + " ".join(f"{v:+.4f}" for v in out), flush=True)
