#!/bin/bash
# round 4, run 50: EXPERIMENT -- W pre-split into planes for the 256x128 split kernel (no VALU for W, 6 B per weight from L2)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
export AVSEP_LIB=dev
set -o pipefail
AVSEP_SPLIT_WPLANES=2 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split" 2>&1 | tail -3 || exit 1
for e in 0 1 0 1; do echo "== W planes: $e"; if [ $e = 1 ]; then export AVSEP_SPLIT_WPLANES=1; else unset AVSEP_SPLIT_WPLANES; fi; AVSEP_SPLIT_VARIANT=2 timeout -k 10 300 python3 tools/gemm_split_probe.py 2>&1 | grep -v amdgpu.ids | head -9 | cut -c1-24,97-135; done | tee $O/gemm_split_probe_wplanes.txt
one() { timeout -k 10 200 python3 bench.py --no-cpu --no-profile --no-also --no-quality "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for i in 1 2; do for w in cfg3 cfg5; do
  unset AVSEP_SPLIT_WPLANES; echo -n "$w W split in the kernel : "; one --workload $w --steps 20 --warmup 3 --rounds 5
  export AVSEP_SPLIT_WPLANES=1; echo -n "$w W planes from memory  : "; one --workload $w --steps 20 --warmup 3 --rounds 5
done; done 2>&1 | tee $O/ab_split_wplanes_256x128.txt
