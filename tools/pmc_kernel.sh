#!/bin/bash
# Developer tool (GPU box): matrix-pipe / VALU / LDS PMC passes over tools/one_kernel.py (counters in their own runs, no tracing).
#   tools/pmc_kernel.sh <kernel-name-substring> <one_kernel.py arguments...>     e.g. tools/pmc_kernel.sh attention_h2 attn_h2 32 8 501
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; KN=$1; shift
TAG=$(echo "$KN $*" | tr ' <>,' '____'); OUT=$R/gpurun_out/pmc_kernel/$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $OUT/p1 -- python3 $R/tools/one_kernel.py "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/p2 -- python3 $R/tools/one_kernel.py "$@" > /dev/null 2>&1
export KN OUT ARGS="$*"
python3 - <<'PY'
import csv, glob, collections, os
OUT=os.environ["OUT"]; KN=os.environ["KN"]
tot=collections.defaultdict(float); n=collections.Counter()
for p in ("p1","p2"):
    for f in glob.glob(f"{OUT}/{p}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if KN not in r["Kernel_Name"]: continue
            tot[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
print("==", KN, os.environ["ARGS"], "(per launch)")
for k in sorted(tot): print(f"  {k:28s} {tot[k]/max(1,n[k]):16.1f}")
gui=tot["GRBM_GUI_ACTIVE"]/max(1,n["GRBM_GUI_ACTIVE"])/8
if gui:
    mf=tot['SQ_VALU_MFMA_BUSY_CYCLES']/max(1,n['SQ_VALU_MFMA_BUSY_CYCLES'])
    print(f"  gui cycles/launch {gui:.0f};  matrix pipes busy = {mf/(gui*1024)*100:.1f} % of (cycles x 1024 SIMDs);  VALU instructions per MFMA = {tot['SQ_INSTS_VALU']/max(1,n['SQ_INSTS_VALU'])/max(1.0,tot['SQ_INSTS_MFMA']/max(1,n['SQ_INSTS_MFMA'])):.2f};  LDS bank-conflict cycles / active = {tot['SQ_LDS_BANK_CONFLICT']/max(1,tot['SQ_LDS_IDX_ACTIVE'])*100:.1f} %")
PY
