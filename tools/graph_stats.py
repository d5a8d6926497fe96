#!/usr/bin/env python3
"""Developer tool (GPU box): fold a `rocprofv3 --kernel-trace --stats` CSV of the timed region (graph replays only:
`bench.py --no-cpu --no-profile --inflight R --steps K --warmup W --rounds 1`) into profiles-style JSON keyed by the kernel
instance names bench.py's live profiler uses, so bench.py can print every kernel's duration INSIDE the replayed graph
(other kernels beside it on the chip) next to its duration alone on the chip.
    tools/graph_stats.py <kernel_stats.csv> <workload> <replays> <out.json>"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
from av_separation import _native


def norm(name):
    return name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]


path, wl, replays, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
ks = {}
for r in csv.DictReader(open(path)):
    if "anonymous" not in r["Name"]:
        continue
    n = norm(r["Name"])
    if n.startswith(("pack_", "scale_copy", "pe_fill", "void at::", "at::", "split_h2_kernel", "split_planes_kernel", "h2_row_stats_kernel")):
        # weight packing / weight planes / ATen checks: not part of a replayed step
        continue
    ks[n] = {"calls_per_step": int(r["Calls"]) / replays, "avg_us": float(r["AverageNs"]) / 1e3,
             "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
bid = _native.load().avsep_build_id().decode()
doc = {"build_id": bid, "note": "rocprofv3 --kernel-trace --stats of the timed region (hipGraph replays only); names as the live "
       "profiler prints them", "workloads": {}}
if os.path.exists(out):
    old = json.load(open(out))
    if old.get("build_id") == bid:
        doc["workloads"] = old.get("workloads", {})
doc["workloads"][wl] = ks
json.dump(doc, open(out, "w"), indent=1)
print(f"{out}: {len(ks)} kernel instances of {wl}, build {bid}")
