#!/usr/bin/env python3
"""Developer tool: in a rocprofv3 kernel trace of the training bench, which kernels run right after each fill / copy launch?
(attributes the anonymous at::native fills and copies of a step to the op that needs them)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70] for r in rows]
for key in ("FillFunctor", "copyBuffer", "direct_copy"):
    ctx = collections.Counter()
    for i, n in enumerate(names):
        if key in n:
            nxt = [m for m in names[i + 1:i + 3]]
            ctx[" -> ".join(x[:48] for x in nxt)] += 1
    print(f"== {key}: {sum(ctx.values())} launches; the two kernels that follow, by count")
    for k, v in ctx.most_common(14):
        print(f"   {v:5d}  {k}")
