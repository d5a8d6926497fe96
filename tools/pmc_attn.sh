#!/bin/bash
# Developer tool (GPU box): PMC passes over tools/attn_bench.py (attention kernel alone). Counters in their own runs.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_attn
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $OUT/p1 -- python3 $R/tools/attn_bench.py > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p2 -- python3 $R/tools/attn_bench.py > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/p3 -- python3 $R/tools/attn_bench.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, os
OUT=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_attn"
tot=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
for p in ("p1","p2","p3"):
    for f in glob.glob(f"{OUT}/{p}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "attention" not in r["Kernel_Name"]: continue
            key=r["Kernel_Name"].split("(")[0].replace("void (anonymous namespace)::","")+" grid="+r.get("Grid_Size","?")
            tot[key][r["Counter_Name"]]+=float(r["Counter_Value"]); n[key][r["Counter_Name"]]+=1
for key in tot:
    t=tot[key]; c=n[key]
    g=lambda k: t[k]/max(1,c[k])
    print("==", key, "launches", c["SQ_WAVE_CYCLES"])
    wc=g("SQ_WAVE_CYCLES"); gui=g("GRBM_GUI_ACTIVE")/8
    if not wc or not gui: continue
    print(f"  gui cycles {gui:.0f} ({gui/2400:.1f} us at 2.4 GHz); mfma busy {g('SQ_VALU_MFMA_BUSY_CYCLES')/(gui*1024)*100:.1f} %; insts: mfma {g('SQ_INSTS_MFMA'):.0f} valu {g('SQ_INSTS_VALU'):.0f} lds {g('SQ_INSTS_LDS'):.0f} salu {g('SQ_INSTS_SALU'):.0f} vmem_rd {g('SQ_INSTS_VMEM_RD'):.0f}")
    for k in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_LDS","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_SCA","SQ_ACTIVE_INST_MISC"):
        if c[k]: print(f"  {k:20s} / SQ_WAVE_CYCLES = {g(k)/wc*100:6.1f} %")
    print(f"  LDS idx active / (gui x 256) = {g('SQ_LDS_IDX_ACTIVE')/(gui*256)*100:.1f} %; bank conflict / idx active = {t['SQ_LDS_BANK_CONFLICT']/max(1,t['SQ_LDS_IDX_ACTIVE'])*100:.1f} %")
PY
