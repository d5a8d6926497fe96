#!/bin/bash
# round 4: the split-precision GEMM inside the big configs (developer library, AVSEP_GEMM_SPLIT=<min 128x128 tiles>)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04n; mkdir -p $O; cd $R
export AVSEP_LIB=dev
one() { timeout -k 10 200 python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for w in cfg3 cfg5; do
  echo -n "$w fp32 MFMA        : "; one --workload $w --steps 20 --warmup 3 --rounds 5
  echo -n "$w split >= 64 tiles: "; AVSEP_GEMM_SPLIT=64 one --workload $w --steps 20 --warmup 3 --rounds 5
  echo -n "$w split >= 16 tiles: "; AVSEP_GEMM_SPLIT=16 one --workload $w --steps 20 --warmup 3 --rounds 5
done 2>&1 | tee $O/ab_split_big_configs.txt
echo -n "train cfg4 fp32 : "; one --mode train --steps 10 --warmup 3 2>&1 | tee -a $O/ab_split_big_configs.txt
echo -n "train cfg4 split: "; AVSEP_GEMM_SPLIT=16 one --mode train --steps 10 --warmup 3 2>&1 | tee -a $O/ab_split_big_configs.txt
AVSEP_GEMM_SPLIT=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_train_gpu.py -m gpu -q -k "baseline_configs or big_configs or matches_reference or trained_config1 or offset_residual" > $O/parity_with_split.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/parity_with_split.txt
