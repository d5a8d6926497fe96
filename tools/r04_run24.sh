#!/bin/bash
# round 4, run 24: 256x128 split kernel -- shape probe with small-M shapes, whole-model A/B (variant 1 / 2 / product rule), PMC
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04w; mkdir -p $O; cd $R
export AVSEP_LIB=dev
timeout -k 10 300 python3 tools/gemm_split_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_split_probe_v2.txt
one() { timeout -k 10 200 python3 bench.py --no-cpu --no-profile --no-also --no-quality "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for i in 1 2; do for w in cfg3 cfg5; do
  echo -n "$w 128x128 kernel everywhere : "; AVSEP_SPLIT_VARIANT=1 one --workload $w --steps 20 --warmup 3 --rounds 5
  echo -n "$w 256x128 kernel everywhere : "; AVSEP_SPLIT_VARIANT=2 one --workload $w --steps 20 --warmup 3 --rounds 5
  echo -n "$w product rule (>= 192 tiles): "; one --workload $w --steps 20 --warmup 3 --rounds 5
done; done 2>&1 | tee $O/ab_split_256x128.txt
export AVSEP_GEMM_SPLIT=1 AVSEP_SPLIT_VARIANT=2
bash tools/pmc_gemm.sh "16064 2048 512" "16064 512 2048" > $O/pmc_gemm_split_256x128.txt 2>&1; tail -45 $O/pmc_gemm_split_256x128.txt
