#!/bin/bash
# GPU box: per-kernel time of the timed region (graph replays) of one workload, current build.  kstats.sh <workload|train> <steps>
W=${1:-cfg3}; K=${2:-10}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kstats; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
if [[ $W == train ]]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$W -- python3 $R/bench.py --mode train --steps $K --warmup 2 --rounds 1 --no-cpu > $O/$W.json 2>/dev/null
else
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$W -- python3 $R/bench.py --workload $W --no-cpu --no-profile --steps $K --warmup 2 --rounds 1 > $O/$W.json 2>/dev/null
fi
cp $(find $O/kt_$W -name "*kernel_stats.csv") $O/${W}_kernel_stats.csv; rm -rf $O/kt_$W
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/${W}_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("$W total kernel ms", tot/1e6)
for r in rows[:22]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['AverageNs'])/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
