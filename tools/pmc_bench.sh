#!/bin/bash
# Developer tool (GPU box): HBM-side traffic per kernel of the bench forward (separate PMC passes, no tracing).
#   tools/pmc_bench.sh [workload [steps warmup]]     default cfg2 20 5; other workloads are merged into the same JSON
cd /tmp; export TMPDIR=/tmp
WL=${1:-cfg2}; STEPS=${2:-20}; WARM=${3:-5}; export WL STEPS WARM
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_bench; mkdir -p $OUT; rm -rf $OUT/fetch $OUT/write $OUT/l2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --workload $WL --steps $STEPS --warmup $WARM --rounds 1 --no-cpu --no-profile --no-graph --inflight 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --workload $WL --steps $STEPS --warmup $WARM --rounds 1 --no-cpu --no-profile --no-graph --inflight 1 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -- python3 $R/bench.py --workload $WL --steps $STEPS --warmup $WARM --rounds 1 --no-cpu --no-profile --no-graph --inflight 1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]; OUT=R+"/gpurun_out/pmc_bench"
tot=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.Counter()
for tag in ("fetch","write","l2"):
    f=glob.glob(f"{OUT}/{tag}/*/*counter_collection.csv")[0]
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "anonymous" not in n: continue
        n=n.replace("void (anonymous namespace)::","").replace("(anonymous namespace)::","").split("(")[0]
        tot[n][r["Counter_Name"]]+=float(r["Counter_Value"])
        if tag=="fetch": calls[n]+=1
fw=float(os.environ["STEPS"])+float(os.environ["WARM"])   # forwards traced: warm-up + ONE round of steps
print(f"{'kernel':34s} {'calls/fwd':>9s} {'fetch MB/fwd (x2 corr)':>22s} {'write MB/fwd':>12s} {'L2 hit%':>8s}")
TF=TW=0
for n,c in sorted(tot.items(), key=lambda kv: -kv[1].get("FETCH_SIZE",0)):
    f=c.get("FETCH_SIZE",0)*1024*2/1e6/fw; w=c.get("WRITE_SIZE",0)*1024/1e6/fw   # FETCH_SIZE in KB, x2 gfx950 correction
    h=c.get("TCC_HIT_sum",0); m=c.get("TCC_MISS_sum",0)
    TF+=f; TW+=w
    print(f"{n:34s} {calls[n]/fw:9.1f} {f:22.1f} {w:12.1f} {100*h/max(1,h+m):8.1f}")
print(f"TOTAL per forward: fetch {TF:.0f} MB (corrected x2), write {TW:.0f} MB")
import json
js={n:{"calls_per_forward":calls[n]/fw,
       "fetch_bytes_per_launch":c.get("FETCH_SIZE",0)*1024*2/max(1,calls[n]),
       "write_bytes_per_launch":c.get("WRITE_SIZE",0)*1024/max(1,calls[n]),
       "l2_hit_rate":c.get("TCC_HIT_sum",0)/max(1,c.get("TCC_HIT_sum",0)+c.get("TCC_MISS_sum",0))} for n,c in tot.items()}
import subprocess, sys
bid=subprocess.run([sys.executable,"-c","import sys;sys.path.insert(0,'%s/av-separation-transformer_amd');from av_separation import _native;print(_native.load().avsep_build_id().decode())"%R],capture_output=True,text=True).stdout.strip()
path=OUT+"/pmc_hbm_traffic.json"
doc={"build_id":bid,"note":"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum in separate passes over `bench.py --no-graph`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced stream); per-launch figures are averages over all launches of the instance in one forward of the workload","kernels":{},"workloads":{}}
if os.path.exists(path):
    old=json.load(open(path))
    if old.get("build_id")==bid: doc["kernels"]=old.get("kernels",{}); doc["workloads"]=old.get("workloads",{})
wl=os.environ["WL"]
doc["workloads"][wl]=js
if wl=="cfg2": doc["kernels"]=js          # the headline workload stays where round 1/2 readers look for it
json.dump(doc, open(path,"w"), indent=1)
PY
