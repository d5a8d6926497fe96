#!/bin/bash
# Developer tool (GPU box): the clock the chip holds under this repo's fp32-MFMA GEMM on random data.
# GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / dispatch duration on >= 9 ms dispatches of a 65536 x 4096 x 2048
# problem (MI355X_MICROARCH.md "DVFS give-back": within 3 % of the in-kernel clock on dispatches of 10 ms or more),
# after ~1 s of back-to-back launches.  Also prints matrix-pipe busy %.  Usage: tools/clock_probe.sh [TILE ...]
export AVSEP_LIB=dev   # developer switches exist only in libavsep_hip_dev.so (make dev)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/clock_probe; rm -rf $OUT; mkdir -p $OUT
TILES=${@:-"auto 128x128x32"}
for t in $TILES; do
  if [ "$t" != auto ]; then export AVSEP_GEMM_TILE=$t; else unset AVSEP_GEMM_TILE; fi
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/$t -- python3 $R/tools/gemm_one.py 65536 4096 2048 150 > /dev/null 2>&1
  python3 - "$OUT/$t" "$t" <<'PY'
import csv, glob, sys, statistics
d, tile = sys.argv[1], sys.argv[2]
cc = glob.glob(d + "/*/*counter_collection.csv"); kt = glob.glob(d + "/*/*kernel_trace.csv")
if not cc or not kt:
    print(tile, "no profiler output"); sys.exit(0)
dur = {}
for r in csv.DictReader(open(kt[0])):
    if "gemm" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
gui, busy = {}, {}
for r in csv.DictReader(open(cc[0])):
    if r["Dispatch_Id"] in dur:
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": gui[r["Dispatch_Id"]] = float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES": busy[r["Dispatch_Id"]] = float(r["Counter_Value"])
ids = sorted(gui, key=int)[len(gui) // 2:]          # second half: the clock has settled
ghz = [gui[i] / 8.0 / dur[i][0] for i in ids]
ms = [dur[i][0] / 1e6 for i in ids]
mf = [busy[i] / 1024.0 / (gui[i] / 8.0) for i in ids if i in busy]
flop = 2.0 * 65536 * 4096 * 2048
print(f"{tile:>12s} {dur[ids[0]][1].split('(')[0][-40:]:40s} {statistics.median(ms):7.3f} ms  {flop / statistics.median(ms) / 1e9:6.1f} TFLOP/s  "
      f"clock {statistics.median(ghz):.3f} GHz  (fp32 matrix peak at that clock {statistics.median(ghz) * 65.536:.1f} TFLOP/s)  "
      f"matrix pipes busy {100 * statistics.median(mf):.1f} %")
PY
done
