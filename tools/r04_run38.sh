#!/bin/bash
# round 4, run 38: small batches of the d_model = 512 configurations -- split-precision kernels against the fp32 MFMA build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
export AVSEP_LIB=dev
one() { timeout -k 10 200 python3 bench.py --no-cpu --no-profile --no-also --no-quality "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for w in cfg3 cfg5; do for b in 1 2 4 8 16; do
  echo -n "$w batch $b split-precision GEMM + attention : "; one --workload $w --batch $b --steps 20 --warmup 3 --rounds 5
  echo -n "$w batch $b fp32 MFMA everywhere             : "; AVSEP_NO_SPLIT=1 one --workload $w --batch $b --steps 20 --warmup 3 --rounds 5
done; done 2>&1 | tee $O/ab_split_small_batches.txt
