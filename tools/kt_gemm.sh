#!/bin/bash
# Developer tool (GPU box): rocprof kernel durations of single GEMM shapes for several library variants.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "" alt4 alt2; do
  if [ -n "$v" ]; then export AVSEP_LIB=$R/av-separation-transformer_amd/lib/$v/libavsep_hip.so; else unset AVSEP_LIB; fi
  for shape in "2016 256 256" "2016 256 1024" "2016 1024 256" "2016 768 256"; do
    tag=$(echo $shape | tr ' ' 'x')
    rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/tools/gemm_one.py $shape 30 > /dev/null 2>&1
    f=$(find /tmp/kt -name "*kernel_stats.csv")
    echo "lib=${v:-ring8} $tag $(grep gemm_kernel $f | awk -F, '{print $1, "avg_ns", $4, "min", $6}' | cut -c30-120)"
  done
done
