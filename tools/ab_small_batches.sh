#!/bin/bash
# Developer tool (GPU box): small batches of the d_model = 512 configurations -- the default split-precision path (two fp16 terms /
# three bf16 terms) against the same library with the fp32 MFMA kernels (AVSEP_NO_SPLIT, what split_precision=False selects).
#   tools/ab_small_batches.sh > gpurun_out/rNN_ab_small_batches.txt
R=$GRAFT_REPO_ROOT; cd $R
export AVSEP_LIB=dev
one() { timeout -k 10 200 python3 bench.py --no-cpu --no-profile --no-also --no-quality "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
echo "# clips/s (two steps in flight) | ms per step (two in flight) | ms per step (one at a time)"
for w in cfg3 cfg5; do for b in 1 2 4 8 16; do
  echo -n "$w batch $b split precision (default) : "; one --workload $w --batch $b --steps 20 --warmup 3 --rounds 5
  echo -n "$w batch $b fp32 MFMA kernels         : "; AVSEP_NO_SPLIT=1 one --workload $w --batch $b --steps 20 --warmup 3 --rounds 5
done; done
