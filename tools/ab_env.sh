#!/bin/bash
# Developer tool (GPU box): A/B one developer environment switch on ONE device in ONE run, alternating.
#   tools/ab_env.sh VAR [bench.py args...]        e.g. tools/ab_env.sh AVSEP_EPI_LATE --steps 200
export AVSEP_LIB=dev   # developer switches exist only in libavsep_hip_dev.so (make dev)
R=$GRAFT_REPO_ROOT; V=$1; shift
for i in 1 2 3; do
  echo -n "$V unset: "; python3 $R/bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  echo -n "$V=1    : "; env $V=1 python3 $R/bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
