#!/bin/bash
# round 4, run 45: split-precision weight gradients with other slice counts (developer library switch; applies to every wgrad launch)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04t; mkdir -p $O; cd $R
export AVSEP_LIB=dev
one() { timeout -k 10 300 python3 bench.py --mode train --workload cfg4 --no-cpu --no-profile --no-also --no-quality "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for sl in plan 16 8 4 2; do
  if [ $sl = plan ]; then unset AVSEP_WGRAD_SLICES; else export AVSEP_WGRAD_SLICES=$sl; fi
  echo -n "slices $sl: split wgrad : "; one --steps 10 --warmup 3 --rounds 3
  echo -n "slices $sl: fp32 wgrad  : "; one --steps 10 --warmup 3 --rounds 3 --train-fp32-wgrad
done 2>&1 | tee $O/ab_train_split_wgrad_slices.txt
