#!/bin/bash
# round 3: did anything of this round slow the LARGE configs with two steps in flight?  same box, alternating libraries
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03ag; mkdir -p $O; cd $R
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing'].get('ms_per_step_min'), d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
L=$R/av-separation-transformer_amd/lib
for wl in cfg3 cfg5; do
for i in 1 2 3; do
  echo -n "$wl build cd8547e8 (before the VALU trims): "; AVSEP_LIB=$L/libavsep_hip_cd85.so one --workload $wl --steps 20 --rounds 5
  echo -n "$wl current                               : "; one --workload $wl --steps 20 --rounds 5
done
done > $O/big_ab.txt 2>&1
cat $O/big_ab.txt
