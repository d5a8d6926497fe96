#!/usr/bin/env python3
"""Developer tool (GPU box): one forward under schedule 1 on a small batch, timed, with the chained launches' status; a
watchdog thread peeks at the plans' state words if the forward does not come back."""
import ctypes as C, faulthandler, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "av-separation-transformer_amd")):
    sys.path.insert(0, p)
import torch
import av_separation as av
from av_separation import _native
faulthandler.dump_traceback_later(40, exit=False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
group = int(sys.argv[2]) if len(sys.argv) > 2 else 8
skew = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
sched = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda:0")
torch.manual_seed(3)
m = av.AVSeparationTransformer(dropout=0.0).to(dev).eval()
ds = av.SyntheticAVDataset(num_samples=B)
items = [ds[i] for i in range(B)]
mixed = torch.stack([x["mixed_spec"] for x in items]).to(dev)
lips = torch.stack([x["lip_frames"] for x in items]).to(dev)
print("model built", flush=True)
done = threading.Event()


def watchdog():
    if done.wait(15):
        return
    lib = _native.load()
    buf = (C.c_uint32 * 64)()
    for idx in range(2):
        n = lib.avsep_chain_peek(m._engine.ctx, idx, buf, 64)
        print(f"[watchdog] plan {idx}: rc {n}: head {buf[0]} err {buf[1]} counters {list(buf[4:40])}", flush=True)


with torch.no_grad():
    sep0, masks0 = m(mixed, lips)
    torch.cuda.synchronize()
    print("schedule 0 forward done", flush=True)
    m.set_schedule(sched, group, skew)
    print("set_schedule done", flush=True)
    threading.Thread(target=watchdog, daemon=True).start()
    for i in range(3):
        t0 = time.perf_counter()
        sep1, masks1 = m(mixed, lips)
        print("enqueued", flush=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        done.set()
        try:
            m.chain_status()
            st = "ok"
        except RuntimeError as e:
            st = str(e)
        print(f"B={B} forward {i}: {dt * 1e3:.2f} ms, status {st}, masks equal {torch.equal(masks1, masks0)}, "
              f"max diff {float((masks1 - masks0).abs().max()):.3e}", flush=True)
