#!/bin/bash
# round 4, run 30: split-precision GEMM in the training step -- step A/B (the gradient gates are run 28's)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04t; mkdir -p $O; cd $R
one() { timeout -k 10 300 python3 bench.py --mode train --workload cfg4 --no-cpu --no-profile --no-also --no-quality "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['loss_first_last'])"; }
for i in 1 2; do
  echo -n "cfg4 training, split-precision GEMM (N, K >= 512): "; one --steps 10 --warmup 3 --rounds 5 --train-split-gemm
  echo -n "cfg4 training, fp32 MFMA GEMM everywhere         : "; one --steps 10 --warmup 3 --rounds 5
done 2>&1 | tee $O/ab_train_split_gemm.txt
