#!/bin/bash
# round 4, run 41: 256x128 split kernel with the barrier behind the fourth product (next chunk's first fragments read under the last two)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
export AVSEP_LIB=dev
set -o pipefail
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split" 2>&1 | tail -3 || exit 1
AVSEP_SPLIT_VARIANT=2 timeout -k 10 300 python3 tools/gemm_split_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-24,27-60,97-135,175-200 | tee $O/gemm_split_probe_early_barrier.txt
one() { timeout -k 10 200 python3 bench.py --no-cpu --no-profile --no-also --no-quality "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for i in 1 2; do for w in cfg3 cfg5; do
  echo -n "$w : "; one --workload $w --steps 20 --warmup 3 --rounds 5
done; done 2>&1 | tee $O/bench_early_barrier.txt
