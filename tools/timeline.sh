#!/bin/bash
# GPU box: the kernel timeline of one step (one step at a time), current PRODUCT build.  timeline.sh [workload [batch]]
W=${1:-cfg2}; B=${2:-0}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/timeline; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --workload $W --batch $B --no-cpu --no-profile --no-also --inflight 1 --steps 50 --rounds 1 > /dev/null 2>&1
python3 $R/tools/trace_step.py $(find $O/kt -name "*kernel_trace.csv") > $O/${W}_step_timeline.txt 2>&1; rm -rf $O/kt
cat $O/${W}_step_timeline.txt
