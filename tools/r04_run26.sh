#!/bin/bash
# round 4, run 26: 64x64 split kernel -- precision tests with each variant forced, shape probe, model A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04y; mkdir -p $O; cd $R
export AVSEP_LIB=dev
set -o pipefail
for v in 0 1 2 3; do AVSEP_SPLIT_VARIANT=$v timeout -k 10 240 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split" 2>&1 | tail -3 || exit 1; done
timeout -k 10 400 python3 tools/gemm_split_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_split_probe_3kernels.txt
one() { timeout -k 10 200 python3 bench.py --no-cpu --no-profile --no-also --no-quality "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for i in 1 2; do for w in cfg3 cfg5; do
  echo -n "$w product rule (256x128 from 128 tiles, else 64x64) : "; one --workload $w --steps 20 --warmup 3 --rounds 5
  echo -n "$w 256x128 everywhere                              : "; AVSEP_SPLIT_VARIANT=2 one --workload $w --steps 20 --warmup 3 --rounds 5
  echo -n "$w 64x64 everywhere                                : "; AVSEP_SPLIT_VARIANT=3 one --workload $w --steps 20 --warmup 3 --rounds 5
done; done 2>&1 | tee $O/ab_split_64x64.txt
