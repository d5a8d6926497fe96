#!/bin/bash
# round 4, run 37: hypothesis check -- cfg2 (d_model 256) with its GEMMs on the split-precision kernels and more forwards in flight
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
export AVSEP_LIB=dev
one() { timeout -k 10 200 python3 bench.py --no-cpu --no-profile --no-also --no-quality "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for f in 2 4 8; do
  echo -n "cfg2 product, $f in flight                                      : "; one --steps 50 --warmup 5 --rounds 5 --inflight $f
  echo -n "cfg2 LayerNorm launches + fp32 GEMMs, $f in flight               : "; AVSEP_NO_LNX=1 one --steps 50 --warmup 5 --rounds 5 --inflight $f
  echo -n "cfg2 LayerNorm launches + split GEMMs (64x64 kernel), $f in flight : "; AVSEP_NO_LNX=1 AVSEP_GEMM_SPLIT=1 AVSEP_SPLIT_VARIANT=3 one --steps 50 --warmup 5 --rounds 5 --inflight $f
  echo -n "cfg2 LayerNorm launches + split GEMMs (256x128 kernel), $f in flight: "; AVSEP_NO_LNX=1 AVSEP_GEMM_SPLIT=1 AVSEP_SPLIT_VARIANT=2 one --steps 50 --warmup 5 --rounds 5 --inflight $f
done 2>&1 | tee $O/cfg2_split_gemm_hypothesis.txt
