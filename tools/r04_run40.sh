#!/bin/bash
# round 4, run 40: 64x64 split kernel with a four-deep register ring -- tests, probe, small-batch forwards
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
export AVSEP_LIB=dev
set -o pipefail
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split" 2>&1 | tail -3 || exit 1
echo "== ring"; AVSEP_SPLIT_VARIANT=3 timeout -k 10 300 python3 tools/gemm_split_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-24,139-175 | tee $O/gemm_split_probe_ring.txt
echo "== no ring"; AVSEP_SPLIT_RING=0 timeout -k 10 300 python3 tools/gemm_split_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-24,139-175 | tee $O/gemm_split_probe_noring.txt
one() { timeout -k 10 200 python3 bench.py --no-cpu --no-profile --no-also --no-quality "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for w in cfg3 cfg5; do for b in 1 4 0; do
  echo -n "$w batch $b (0 = benchmark batch) ring     : "; one --workload $w --batch $b --steps 20 --warmup 3 --rounds 5
  echo -n "$w batch $b (0 = benchmark batch) no ring  : "; AVSEP_SPLIT_RING=0 one --workload $w --batch $b --steps 20 --warmup 3 --rounds 5
  echo -n "$w batch $b (0 = benchmark batch) fp32 MFMA: "; AVSEP_NO_SPLIT=1 one --workload $w --batch $b --steps 20 --warmup 3 --rounds 5
done; done 2>&1 | tee $O/ab_split_ring_small_batches.txt
