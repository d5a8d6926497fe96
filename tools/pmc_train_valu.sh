#!/bin/bash
# Developer tool (GPU box): VALU instructions beside the MFMAs, per kernel of the cfg4 TRAINING step (one PMC pass, no tracing)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_train; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 $R/bench.py --mode train --steps 2 --warmup 1 --rounds 1 --no-cpu > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]; OUT=R+"/gpurun_out/pmc_train"
tot=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.Counter()
for f in glob.glob(f"{OUT}/a/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].replace("void (anonymous namespace)::","").replace("(anonymous namespace)::","").split("(")[0][:60]
        tot[n][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Counter_Name"]=="GRBM_GUI_ACTIVE": calls[n]+=1
steps=3.0
print("cfg4 training, per STEP (3 steps profiled): launches, kernel time by GRBM_GUI_ACTIVE/8 at 2.1 GHz (reads high on short dispatches),")
print("matrix-pipe busy time, and the floor of the VALU time beside it = (SQ_INSTS_VALU - SQ_INSTS_MFMA) x 4 cycles / 1024 SIMDs")
print(f"{'kernel':60s} {'launches':>8s} {'gui_us':>9s} {'mfma_us':>9s} {'valu_us>=':>9s} {'valu/mfma':>9s}")
rows=[]
for n,c in tot.items():
    mf=c.get("SQ_INSTS_MFMA",0); va=c.get("SQ_INSTS_VALU",0)-mf
    rows.append((va*4/1024/2100.0/steps, n, calls[n]/steps, c.get("GRBM_GUI_ACTIVE",0)/8/2100.0/steps, c.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/1024/2100.0/steps, va/max(1.0,mf)))
tv=sum(r[0] for r in rows); tm=sum(r[4] for r in rows)
for v,n,k,g,m,ratio in sorted(rows, reverse=True)[:28]:
    print(f"{n:60s} {k:8.1f} {g:9.1f} {m:9.1f} {v:9.1f} {ratio:9.2f}")
print(f"TOTAL per step: matrix-pipe busy {tm:.0f} us, VALU floor {tv:.0f} us")
PY
