#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: kernels per HW queue, busy time per queue, overlap factor."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']) for r in rows)
skip = int(len(ev) * 0.3); ev = ev[skip:len(ev) - skip // 3]       # middle part = steady-state replays
q = collections.Counter(); busy = collections.Counter()
for s, e, qi, n in ev: q[qi] += 1; busy[qi] += e - s
span = max(e for s, e, _, _ in ev) - min(s for s, e, _, _ in ev)
print("span us", span / 1e3, "kernels", len(ev))
for qi in q: print(f"queue {qi}: {q[qi]} kernels, busy {busy[qi]/1e3:.0f} us ({busy[qi]/span:.2f} of span)")
print("sum busy / span =", sum(busy.values()) / span)
