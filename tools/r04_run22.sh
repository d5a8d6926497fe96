#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04v; mkdir -p $O; cd $R
export AVSEP_LIB=dev
one() { timeout -k 10 200 python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for i in 1 2; do for w in cfg3 cfg5; do
  echo -n "$w split: Linear + Conv1d + mask head : "; one --workload $w --steps 20 --warmup 3 --rounds 5
  echo -n "$w split: Linear + Conv1d             : "; AVSEP_SPLIT_NO_MASK=1 one --workload $w --steps 20 --warmup 3 --rounds 5
  echo -n "$w split: Linear + mask head          : "; AVSEP_SPLIT_NO_TAPS=1 one --workload $w --steps 20 --warmup 3 --rounds 5
  echo -n "$w split: Linear only                 : "; AVSEP_SPLIT_NO_TAPS=1 AVSEP_SPLIT_NO_MASK=1 one --workload $w --steps 20 --warmup 3 --rounds 5
done; done 2>&1 | tee $O/ab_split_conv_mask.txt
