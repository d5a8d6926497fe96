"""Dev probe: is a clip's output inside a batch bit-equal to the clip run alone (fuzz-wide cases), repeated, so that
run-to-run variation (a race) shows next to a systematic difference.  Use with AVSEP_LIB=dev and the developer
switches (AVSEP_TAIL_SPLIT=0, AVSEP_NO_H2=1, AVSEP_NO_PLANES=1, AVSEP_SERIAL=1) to find the stage.
Usage: python tools/batch_invariance_probe.py <fuzz-wide case> [...]"""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "av-separation-transformer_amd"))
import torch
import test_fuzz_gpu as f
from oracle import seeded

dev = torch.device("cuda:0")
for case in map(int, sys.argv[1:]):
    cfg, dm = f._draw_wide(random.Random(31000 + case))
    m = f._model(dev, cfg, 300 + case).eval()
    mixed, lips = seeded.inputs(1500 + case, dm["B"], cfg["freq_bins"], dm["T"], dm["N"], dm["H"], dm["W"])
    x, l = torch.from_numpy(mixed).to(dev), torch.from_numpy(lips).to(dev)
    B = dm["B"]
    out = []
    with torch.no_grad():
        ref = [m(x[b:b + 1], l[b:b + 1])[1] for b in range(B)]
        for rep in range(4):
            mk = m(x, l)[1]
            bad = [b for b in range(B) if not torch.equal(mk[b:b + 1], ref[b])]
            worst = max([(mk[b:b + 1] - ref[b]).abs().max().item() for b in range(B)])
            out.append((bad, worst))
        again = [b for b in range(B) if not torch.equal(m(x[b:b + 1], l[b:b + 1])[1], ref[b])]
    print(case, {k: os.environ[k] for k in os.environ if k.startswith("AVSEP_")}, "clips that differ per repeat:", out, "alone twice differs:", again)
