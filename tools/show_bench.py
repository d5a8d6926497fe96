#!/usr/bin/env python3
"""Pretty-print the JSON line of bench.py (stdin or file)."""
import json, sys
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
d = json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])
print(f"value {d['value']} {d['unit']}  ms/step {d['ms_per_step']}  n_gpus {d['n_gpus']}")
print("roofline", d["roofline"])
tot = 0.0
for k in d.get("kernels", []):
    t = k["avg_us"] * k["calls_per_step"]; tot += t
    print(f"  {k['name']:28s} x{k['calls_per_step']:4.0f} avg {k['avg_us']:7.2f} us  tot {t:7.1f}  {k['tflops']:6.1f} TF {k['gbs']:7.1f} GB/s")
print(f"  sum of kernel time per step: {tot:.1f} us")
if "cpu_baseline" in d: print("cpu_baseline", d["cpu_baseline"])
