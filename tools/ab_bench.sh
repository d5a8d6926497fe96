#!/bin/bash
# Developer tool (GPU box): A/B two builds of the library in ONE run on ONE device (cross-run comparisons are worthless
# at the few-% level: devices differ).  lib/libavsep_base.so = the build to compare against (copy it there by hand).
#   tools/ab_bench.sh [bench.py args...]
R=$GRAFT_REPO_ROOT; B=$R/av-separation-transformer_amd/lib/libavsep_base.so
for i in 1 2 3; do
  echo -n "base: "; AVSEP_LIB=$B python3 $R/bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  echo -n "new : "; python3 $R/bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
