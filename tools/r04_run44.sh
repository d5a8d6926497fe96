#!/bin/bash
# round 4, run 44: split-precision weight-gradient kernel -- op test, training tests (strict gates), step A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04t; mkdir -p $O; cd $R
set -o pipefail
timeout -k 10 400 python3 -m pytest tests/test_train_ops_gpu.py -x -q -m gpu -k "wgrad" 2>&1 | tail -8 || exit 1
timeout -k 10 900 python3 -m pytest tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -4 || exit 1
cp gpurun_out/grad_gate_train_cfg4.txt $O/grad_gate_train_cfg4_split_wgrad.txt
one() { timeout -k 10 300 python3 bench.py --mode train --workload cfg4 --no-cpu --no-profile --no-also --no-quality "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['loss_first_last'])"; }
for i in 1 2; do
  echo -n "cfg4 training, dgrad + wgrad split-precision (default): "; one --steps 10 --warmup 3 --rounds 5
  echo -n "cfg4 training, dgrad split-precision, wgrad fp32 MFMA : "; one --steps 10 --warmup 3 --rounds 5 --train-fp32-wgrad
  echo -n "cfg4 training, fp32 MFMA everywhere                   : "; one --steps 10 --warmup 3 --rounds 5 --train-fp32-wgrad --train-fp32-dgrad
  echo -n "cfg4 training, forward + dgrad + wgrad split-precision: "; one --steps 10 --warmup 3 --rounds 5 --train-split-gemm
done 2>&1 | tee $O/ab_train_split_wgrad.txt
