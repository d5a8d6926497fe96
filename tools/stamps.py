#!/usr/bin/env python3
"""Developer tool (GPU box): profiler-free timeline of one cfg2 step under graph replay, from device wall-clock
stamps at the stage boundaries (AVSEP_STAMPS, avsep_read_stamps).  Prints medians over many steps relative to the
step's first stamp, plus the period between consecutive steps."""
import os
os.environ.setdefault("AVSEP_LIB", "dev")   # developer switches live in libavsep_hip_dev.so only
import ctypes as C
import os
import sys

os.environ["AVSEP_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "av-separation-transformer_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import av_separation as av  # noqa: E402
from av_separation import _native  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
wl = dict(wl, model=dict(wl["model"]))
for key, env in (("num_encoder_layers", "AVSEP_LE"), ("num_fusion_layers", "AVSEP_LF")):   # ablation knobs
    if env in os.environ:
        wl["model"][key] = int(os.environ[env])
B = wl["batch"]
torch.manual_seed(0)
m = av.AVSeparationTransformer(dropout=0.0, **wl["model"]).to(dev).eval()
ds = av.SyntheticAVDataset(num_samples=B, **wl["data"])
items = [ds[i] for i in range(B)]
mixed = torch.stack([it["mixed_spec"] for it in items]).to(dev).contiguous()
lips = torch.stack([it["lip_frames"] for it in items]).to(dev).contiguous()
_, F, T = mixed.shape
S = wl["model"]["num_speakers"]
mk, sp = torch.empty(B, T, S, F, device=dev), torch.empty(B, T, S, F, device=dev)
lib = _native.load()
if os.environ.get("AVSEP_SCHEDULE") == "fork":      # the two-stream schedule of rounds 1-2
    names = ["audio start", "visual start", "visual enc done", "kv proj done", "audio done", "tail start (main)",
             "tail start (side)", "tail end (main)", "tail end (side)", "step end"]
else:                                               # paired schedule (forward_paired): slots 6 and 8 are not stamped
    names = ["audio front start", "visual front start", "visual front done (conv stack + frame proj)", "resize + K/V projection done",
             "audio front done", "encoder layers done (paired launches)", None, "fusion layers done", None, "step end (decoder done)"]
st = torch.cuda.Stream(device=dev)
rows = []
with torch.cuda.stream(st), torch.no_grad():
    for _ in range(20):
        m.run_static(mixed, lips, mk, sp, graph=True)
    st.synchronize()
    for it in range(40):
        # a burst keeps the queue full; the stamps that survive are the LAST step's
        for _ in range(8):
            m.run_static(mixed, lips, mk, sp, graph=True)
        st.synchronize()
        buf = (C.c_uint64 * 10)()
        _native.check(lib.avsep_read_stamps(m._engine.ctx, buf, 10), "read_stamps")
        rows.append([int(x) for x in buf])
a = np.array(rows, dtype=np.float64) / 100.0           # us
used = [i for i, n in enumerate(names) if n is not None]
a = a[:, used]
t0 = a.min(axis=1, keepdims=True)
rel = np.median(a - t0, axis=0)
for i in np.argsort(rel):
    print(f"{rel[i]:8.1f} us  {names[used[i]]}")
