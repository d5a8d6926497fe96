#!/bin/bash
# round 4: A/B of a scheduling priority for the visual branch's stream (developer library, AVSEP_SIDE_PRIO)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04j; mkdir -p $O; cd $R
export AVSEP_LIB=dev
one() { timeout -k 10 100 python3 bench.py --no-cpu --no-profile --steps 200 --warmup 20 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for i in 1 2 3; do
  echo -n "default priority : "; one
  echo -n "visual HIGH      : "; AVSEP_SIDE_PRIO=high one
  echo -n "visual LOW       : "; AVSEP_SIDE_PRIO=low one
done 2>&1 | tee $O/ab_side_priority.txt
for w in cfg3 cfg5; do
  echo -n "$w default : "; one --workload $w --steps 20 --warmup 3 --rounds 5
  echo -n "$w HIGH    : "; AVSEP_SIDE_PRIO=high one --workload $w --steps 20 --warmup 3 --rounds 5
done 2>&1 | tee -a $O/ab_side_priority.txt
