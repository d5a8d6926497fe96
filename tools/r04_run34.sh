#!/bin/bash
# round 4, run 34: split attention without clamps; 2 vs 3 workgroups per CU
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O; cd $R
set -o pipefail
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "attention_split" 2>&1 | tail -4 || exit 1
AVSEP_LIB=dev AVSEP_ATTN_SPLIT_WGS=3 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "attention_split" 2>&1 | tail -4 || exit 1
echo "== 2 workgroups per CU"; timeout -k 10 300 python3 tools/attention_split_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/attention_split_probe_2wg.txt
echo "== 3 workgroups per CU"; AVSEP_ATTN_SPLIT_WGS=3 timeout -k 10 300 python3 tools/attention_split_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/attention_split_probe_3wg.txt
export AVSEP_LIB=dev
one() { timeout -k 10 200 python3 bench.py --no-cpu --no-profile --no-also --no-quality "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for i in 1 2; do for w in cfg3 cfg5; do
  echo -n "$w split attention, 2 workgroups per CU : "; one --workload $w --steps 20 --warmup 3 --rounds 5
  echo -n "$w split attention, 3 workgroups per CU : "; AVSEP_ATTN_SPLIT_WGS=3 one --workload $w --steps 20 --warmup 3 --rounds 5
done; done 2>&1 | tee $O/ab_split_attention_wgs.txt
