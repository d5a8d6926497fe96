#!/usr/bin/env python3
"""Print the two-queue timeline of ONE graph-replay step from a rocprofv3 kernel_trace.csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']) for r in rows)
starts = [i for i, e in enumerate(ev) if 'transpose_pad' in e[3]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
seg = ev[starts[k]:starts[k + 1]]
t0 = seg[0][0]
qs = sorted(set(e[2] for e in seg))
print(f"step span {(max(e[1] for e in seg) - t0) / 1e3:.1f} us, {len(seg)} kernels, queues {qs}")
for e in seg:
    name = e[3].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:34]
    col = qs.index(e[2])
    print(f"{(e[0]-t0)/1e3:8.1f} {(e[1]-t0)/1e3:8.1f} {(e[1]-e[0])/1e3:7.1f}  " + " " * (38 * col) + name)
