#!/bin/bash
# round 4, run 42: 256x128 split kernel -- the split spread evenly over the six products (22 VALU behind each 16 MFMAs) vs 44 behind products 2, 4, 6
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
export AVSEP_LIB=dev
set -o pipefail
AVSEP_SPLIT_EVEN=1 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split" 2>&1 | tail -3 || exit 1
for e in 0 1 0 1; do echo "== AVSEP_SPLIT_EVEN=$e"; AVSEP_SPLIT_EVEN=$e AVSEP_SPLIT_VARIANT=2 timeout -k 10 300 python3 tools/gemm_split_probe.py 2>&1 | grep -v amdgpu.ids | head -9 | cut -c1-24,97-135; done | tee $O/gemm_split_probe_even.txt
one() { timeout -k 10 200 python3 bench.py --no-cpu --no-profile --no-also --no-quality "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for i in 1 2; do for w in cfg3 cfg5; do
  echo -n "$w split behind products 2, 4, 6 : "; AVSEP_SPLIT_EVEN=0 one --workload $w --steps 20 --warmup 3 --rounds 5
  echo -n "$w split spread over all six     : "; AVSEP_SPLIT_EVEN=1 one --workload $w --steps 20 --warmup 3 --rounds 5
done; done 2>&1 | tee $O/ab_split_even.txt
