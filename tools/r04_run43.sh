#!/bin/bash
# round 4, run 43: training with the activation-gradient GEMMs on the split-precision kernels -- strict gradient gates, step A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04t; mkdir -p $O; cd $R
set -o pipefail
timeout -k 10 900 python3 -m pytest tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -6 || exit 1
cp gpurun_out/grad_gate_train_cfg4.txt $O/grad_gate_train_cfg4_split_dgrad.txt
one() { timeout -k 10 300 python3 bench.py --mode train --workload cfg4 --no-cpu --no-profile --no-also --no-quality "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['loss_first_last'])"; }
for i in 1 2; do
  echo -n "cfg4 training, dgrad GEMMs split-precision (default): "; one --steps 10 --warmup 3 --rounds 5
  echo -n "cfg4 training, fp32 MFMA GEMM everywhere            : "; one --steps 10 --warmup 3 --rounds 5 --train-fp32-dgrad
  echo -n "cfg4 training, forward + dgrad split-precision      : "; one --steps 10 --warmup 3 --rounds 5 --train-split-gemm
done 2>&1 | tee $O/ab_train_split_dgrad.txt
