#!/bin/bash
# Developer tool (GPU box): matrix-core utilisation and wave-stall split per kernel of the bench forward
# (PMC passes only, no tracing; eager launches so every kernel is its own dispatch record).
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_mfma; rm -rf $OUT; mkdir -p $OUT
ARGS="$R/bench.py --steps 20 --warmup 5 --no-cpu --no-profile --rounds 1 --no-graph --inflight 1 ${BENCH_EXTRA}"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc MfmaUtil --output-format csv -d $OUT/c -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/d -- python3 $ARGS > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]; OUT=R+"/gpurun_out/pmc_mfma"
tot=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.Counter()
for tag in ("a","b","c","d"):
    fs=glob.glob(f"{OUT}/{tag}/*/*counter_collection.csv")
    if not fs: continue
    seen=set()
    for r in csv.DictReader(open(fs[0])):
        n=r["Kernel_Name"]
        if "anonymous" not in n: continue
        n=n.replace("void (anonymous namespace)::","").replace("(anonymous namespace)::","").split("(")[0]
        tot[n][r["Counter_Name"]]+=float(r["Counter_Value"])
        if tag=="a" and r["Counter_Name"]=="GRBM_GUI_ACTIVE": calls[n]+=1
print("per kernel, summed over all its launches of 25 eager forwards (cfg2, B=32); MI355X: 256 CUs x 4 SIMDs")
print("mfma_busy% = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 * 4)  [GRBM_GUI_ACTIVE is summed over the 8 XCDs]")
print("MOPS_F32 x 512 = fp32 MFMA flops (one MOP = 512 flops);  wave-cycle split: wait_any / wait_inst / active (quad-cycles)")
print("mfma_us = SQ_VALU_MFMA_BUSY_CYCLES per launch / 1024 SIMDs / 2.1 GHz: time the matrix pipes were busy; compare with the")
print("kernel's HIP-event duration in the bench JSON (GRBM_GUI_ACTIVE reads high on dispatches this short, so mfma_busy% is a floor)")
print(f"{'kernel':34s} {'launch':>6s} {'gui_us/launch':>13s} {'mfma_us':>8s} {'mfma_busy%':>10s} {'MfmaUtil':>9s} {'GFLOP(mfma)/launch':>18s} {'wait_any%':>9s} {'wait_inst%':>10s} {'active%':>8s}")
for n,c in sorted(tot.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE",0)):
    k=max(1,calls[n]); gui=c.get("GRBM_GUI_ACTIVE",0)/8.0
    busy=100*c.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/max(1.0,gui*256*4)
    wc=max(1.0,c.get("SQ_WAVE_CYCLES",0))
    print(f"{n:34s} {k:6d} {gui/k/2100.0:13.2f} {c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/k/1024/2100.0:8.2f} {busy:10.1f} {c.get('MfmaUtil',0)/k:9.1f} {c.get('SQ_INSTS_VALU_MFMA_MOPS_F32',0)*512/k/1e9:18.4f} "
          f"{100*c.get('SQ_WAIT_ANY',0)/wc:9.1f} {100*c.get('SQ_WAIT_INST_ANY',0)/wc:10.1f} {100*c.get('SQ_ACTIVE_INST_ANY',0)/wc:8.1f}")
print()
print("VALU work beside the MFMAs (fp32 MFMA and VALU do not overlap on a SIMD, profiles/r03_mfma_valu_exclusive.txt): instructions per launch,")
print("valu_us = (SQ_INSTS_VALU - SQ_INSTS_MFMA) x 4 cycles / 1024 SIMDs / 2.1 GHz (a floor: packed / transcendental / DPP instructions take 8-13)")
print(f"{'kernel':34s} {'mfma/launch':>12s} {'other valu':>12s} {'valu per mfma':>13s} {'mfma_us':>8s} {'valu_us >=':>10s} {'lds insts':>10s} {'salu':>10s}")
for n,c in sorted(tot.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE",0)):
    k=max(1,calls[n]); mf=c.get("SQ_INSTS_MFMA",0)/k; va=c.get("SQ_INSTS_VALU",0)/k-mf
    if "SQ_INSTS_VALU" not in c: continue
    print(f"{n:34s} {mf:12.0f} {va:12.0f} {va/max(1.0,mf):13.2f} {c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/k/1024/2100.0:8.2f} {va*4/1024/2100.0:10.2f} {c.get('SQ_INSTS_LDS',0)/k:10.0f} {c.get('SQ_INSTS_SALU',0)/k:10.0f}")
PY
