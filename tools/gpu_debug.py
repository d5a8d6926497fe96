#!/usr/bin/env python3
"""Developer script (GPU box): run the HIP forward on golden fixtures and print per-tap errors."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import av_separation as av
from conftest import load_golden
from helpers import golden_state, golden_inputs, maxabs

names = sys.argv[1:] or ["fwd_tiny", "fwd_odd", "fwd_down", "fwd_t1", "trained_tiny", "fwd_cfg1"]
dev = torch.device("cuda:0")
for name in names:
    g = load_golden(name); c = g["config"]
    m = av.AVSeparationTransformer(c["F"], c["d"], c["h"], c["Le"], c["Lf"], c["S"], dropout=0.0)
    sd = m.state_dict()
    for k, v in golden_state(g).items():
        sd[k] = torch.from_numpy(np.ascontiguousarray(v))
    m.load_state_dict(sd); m.to(dev).eval()
    mixed, lips = golden_inputs(g)
    full = c["full"]
    if full: m.enable_debug_taps(True)
    with torch.no_grad():
        sep, masks = m(torch.from_numpy(mixed).to(dev), torch.from_numpy(lips).to(dev))
    torch.cuda.synchronize()
    print(f"== {name}: masks stride {masks.stride()} range [{masks.min():.4f},{masks.max():.4f}]")
    if full:
        B, T, N, H, W, d = c["B"], c["T"], c["N"], c["H"], c["W"], c["d"]
        for k in sorted(g):
            if not k.startswith("tap."): continue
            tn = k[4:]; ref = g[k]
            if tn in ("a_conv2", "v_proj", "d_logits"): continue
            if tn.startswith("v_conv"):
                Mv, C, h, w = ref.shape
                got = m.read_tap(tn, (Mv, h, w, C)).permute(0, 3, 1, 2)
            else:
                got = m.read_tap(tn, ref.shape)
            print(f"   tap {tn:10s} maxabs {maxabs(got.cpu().numpy(), ref):.3e}  (|ref|max {np.abs(ref).max():.3f})")
        print(f"   masks maxabs {maxabs(masks.cpu().numpy(), g['masks']):.3e}  sep maxabs {maxabs(sep.cpu().numpy(), g['separated']):.3e}")
    else:
        ms = masks.contiguous().cpu().numpy().reshape(-1)[::7]; ss = sep.contiguous().cpu().numpy().reshape(-1)[::7]
        print(f"   masks maxabs {maxabs(ms, g['masks.slice']):.3e}  sep maxabs {maxabs(ss, g['separated.slice']):.3e}")
