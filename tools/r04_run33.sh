#!/bin/bash
# round 4, run 33: split-precision attention -- shape probe and PMC
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/attention_split_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/attention_split_probe.txt
cd /tmp; export TMPDIR=/tmp
P=$O/pmc_attn; rm -rf $P; mkdir -p $P
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $P/p1 -- python3 $R/tools/attention_split_probe.py split > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $P/p2 -- python3 $R/tools/attention_split_probe.py split > /dev/null 2>&1
python3 - <<'PY' | tee $O/pmc_attention_split.txt
import csv, glob, collections, os
P=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04a/pmc_attn"
tot=collections.defaultdict(float); n=collections.Counter()
for f in glob.glob(P+"/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "attention_split" not in r["Kernel_Name"]: continue
        if int(r["Grid_Size"]) != 64*8*2*256 and int(r["Grid_Size"]) != 32*8*4*256: continue
        tot[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
for k in sorted(tot): print(f"  {k:28s} {tot[k]/max(1,n[k]):16.1f}  (n={n[k]})")
gui=tot["GRBM_GUI_ACTIVE"]/max(1,n["GRBM_GUI_ACTIVE"])/8
if gui: print(f"  gui cycles/launch {gui:.0f}; mfma busy {tot['SQ_VALU_MFMA_BUSY_CYCLES']/n['SQ_VALU_MFMA_BUSY_CYCLES']/(gui*1024)*100:.1f} %; LDS idx active {tot['SQ_LDS_IDX_ACTIVE']/n['SQ_LDS_IDX_ACTIVE']/(gui*256)*100:.1f} %; bank conflicts / idx active {tot['SQ_LDS_BANK_CONFLICT']/max(1,tot['SQ_LDS_IDX_ACTIVE'])*100:.1f} %")
wc=tot["SQ_WAVE_CYCLES"]/max(1,n["SQ_WAVE_CYCLES"])
for k in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_VALU"):
    if n[k]: print(f"  {k:20s} / SQ_WAVE_CYCLES = {tot[k]/n[k]/wc*100:6.1f} %")
PY
