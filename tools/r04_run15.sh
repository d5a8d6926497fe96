#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04o; mkdir -p $O; cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "split_precision" 2>&1 | tail -5
timeout -k 10 200 python3 tools/gemm_split_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_split_probe_v2.txt
one() { timeout -k 10 200 python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for w in cfg3 cfg5; do
  echo -n "$w product (split, packed planes) : "; one --workload $w --steps 20 --warmup 3 --rounds 5
  echo -n "$w dev AVSEP_NO_SPLIT (fp32 MFMA) : "; AVSEP_LIB=dev AVSEP_NO_SPLIT=1 one --workload $w --steps 20 --warmup 3 --rounds 5
done 2>&1 | tee $O/ab_split_big_configs_v2.txt
echo -n "cfg2 product: "; one --steps 200 --warmup 20 | tee -a $O/ab_split_big_configs_v2.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/validation.txt 2>&1; echo "pytest rc=$?"; tail -6 $O/validation.txt
