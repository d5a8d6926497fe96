#!/bin/bash
# round 4, run 32: split-precision attention -- op test, big-config parity tests, model A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O; cd $R
set -o pipefail
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "attention_split" 2>&1 | tail -15 || exit 1
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "big_configs or trained or golden or cfg" 2>&1 | tail -5 || exit 1
export AVSEP_LIB=dev
one() { timeout -k 10 200 python3 bench.py --no-cpu --no-profile --no-also --no-quality "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for i in 1 2; do for w in cfg3 cfg5; do
  echo -n "$w split-precision attention : "; one --workload $w --steps 20 --warmup 3 --rounds 5
  echo -n "$w fp32 MFMA attention       : "; AVSEP_NO_SPLIT_ATTN=1 one --workload $w --steps 20 --warmup 3 --rounds 5
done; done 2>&1 | tee $O/ab_split_attention.txt
