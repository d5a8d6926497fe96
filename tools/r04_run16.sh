#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04p; mkdir -p $O; cd $R
export AVSEP_LIB=dev
one() { timeout -k 10 200 python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for i in 1 2; do for w in cfg3 cfg5; do
  echo -n "$w split, W planes packed : "; one --workload $w --steps 20 --warmup 3 --rounds 5
  echo -n "$w split, W split on the fly: "; AVSEP_SPLIT_NO_PLANES=1 one --workload $w --steps 20 --warmup 3 --rounds 5
done; done 2>&1 | tee $O/ab_split_planes.txt
