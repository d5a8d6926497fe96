#!/bin/bash
# Developer tool (GPU box): PMC passes over tools/gemm_planes_one.py.  Counters in their own runs.
#   tools/pmc_planes.sh "<variants>" ["M N K" ...]       e.g. tools/pmc_planes.sh "8 10" "16064 2048 512"
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_planes
rm -rf $OUT; mkdir -p $OUT
VARS=$1; shift
if [ $# -eq 0 ]; then set -- "16064 2048 512"; fi
for v in $VARS; do
export AVSEP_PLANES_V=$v
for shape in "$@"; do
  tag=v${v}_$(echo $shape | tr ' ' 'x')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $OUT/p1_$tag -- python3 $R/tools/gemm_planes_one.py $shape 10 > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p2_$tag -- python3 $R/tools/gemm_planes_one.py $shape 10 > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p3_$tag -- python3 $R/tools/gemm_planes_one.py $shape 10 > /dev/null 2>&1
  rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM SQ_INSTS_WAVE32_LDS SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/p4_$tag -- python3 $R/tools/gemm_planes_one.py $shape 10 > /dev/null 2>&1
done
done
python3 - <<'PY'
import csv, glob, collections, os
OUT=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_planes"
for d in sorted(glob.glob(OUT+"/p1_*")):
    tag=d.split("p1_")[1]
    tot=collections.defaultdict(float); n=collections.Counter()
    for p in ("p1","p2","p3","p4"):
        for f in glob.glob(f"{OUT}/{p}_{tag}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "gemm_planes" not in r["Kernel_Name"]: continue
                tot[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    print("==", tag, "(per launch)")
    for k in sorted(tot): print(f"  {k:28s} {tot[k]/max(1,n[k]):16.1f}")
    wc=tot["SQ_WAVE_CYCLES"]/max(1,n["SQ_WAVE_CYCLES"])
    if wc:
        for k in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_LDS","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_VMEM","SQ_ACTIVE_INST_SCA","SQ_ACTIVE_INST_MISC","SQ_INST_CYCLES_VMEM"):
            if n[k]: print(f"  {k:20s} / SQ_WAVE_CYCLES = {tot[k]/n[k]/wc*100:6.1f} %")
    gui=tot["GRBM_GUI_ACTIVE"]/max(1,n["GRBM_GUI_ACTIVE"])/8
    if gui:
        print(f"  gui cycles/launch {gui:.0f};  mfma busy = {tot['SQ_VALU_MFMA_BUSY_CYCLES']/n['SQ_VALU_MFMA_BUSY_CYCLES']/(gui*1024)*100:.1f} %;  LDS idx active / (gui x 256 CUs) = {tot['SQ_LDS_IDX_ACTIVE']/n['SQ_LDS_IDX_ACTIVE']/(gui*256)*100:.1f} %;  bank conflict cycles / idx active = {tot['SQ_LDS_BANK_CONFLICT']/max(1,tot['SQ_LDS_IDX_ACTIVE'])*100:.1f} %")
PY
