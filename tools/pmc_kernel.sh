#!/bin/bash
# Developer tool (GPU box): SQ/LDS counters for kernels whose name matches $1 during an eager bench run.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=/tmp/pmck; rm -rf $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $OUT/p1 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-profile --no-graph > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/p2 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-profile --no-graph > /dev/null 2>&1
python3 - "$1" <<'PY'
import csv, glob, sys, collections
pat=sys.argv[1]; v=collections.defaultdict(list)
for p in ("p1","p2"):
    for r in csv.DictReader(open(glob.glob(f"/tmp/pmck/{p}/*/*counter_collection.csv")[0])):
        if pat in r["Kernel_Name"]: v[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,x in v.items(): print(f"{k:28s} {sum(x)/len(x):16.0f}   (n={len(x)})")
PY
