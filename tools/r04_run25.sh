#!/bin/bash
# round 4, run 25: persistent 256x128 split kernel -- precision test (persistent and one-tile-per-workgroup grids), probe, model A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04x; mkdir -p $O; cd $R
export AVSEP_LIB=dev
set -o pipefail
timeout -k 10 240 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split" 2>&1 | tail -3 || exit 1
AVSEP_SPLIT_VARIANT=2 timeout -k 10 240 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split" 2>&1 | tail -3 || exit 1
AVSEP_SPLIT_VARIANT=2 AVSEP_SPLIT_GRID=24 timeout -k 10 240 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split" 2>&1 | tail -3 || exit 1
echo "== persistent grid (default)"; timeout -k 10 300 python3 tools/gemm_split_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_split_probe_persistent.txt
echo "== one tile per workgroup"; AVSEP_SPLIT_GRID=0 timeout -k 10 300 python3 tools/gemm_split_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_split_probe_onetile.txt
one() { timeout -k 10 200 python3 bench.py --no-cpu --no-profile --no-also --no-quality "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for i in 1 2; do for w in cfg3 cfg5; do
  echo -n "$w 256x128 persistent (rule: >= 128 tiles) : "; one --workload $w --steps 20 --warmup 3 --rounds 5
  echo -n "$w 256x128 persistent everywhere          : "; AVSEP_SPLIT_VARIANT=2 one --workload $w --steps 20 --warmup 3 --rounds 5
  echo -n "$w 256x128 one tile per workgroup, everywhere: "; AVSEP_SPLIT_VARIANT=2 AVSEP_SPLIT_GRID=0 one --workload $w --steps 20 --warmup 3 --rounds 5
done; done 2>&1 | tee $O/ab_split_persistent.txt
