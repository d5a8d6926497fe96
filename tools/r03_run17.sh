#!/bin/bash
# round 3: LayerNorm-in-the-epilogue GEMM, tile choice inside the step (second sweep)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03t; mkdir -p $O; cd $R
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing'].get('ms_per_step_min'), d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
export AVSEP_LIB=dev
for i in 1 2; do
  echo -n "shipped (LN inside the GEMM prologue): "; one --steps 200 --rounds 5
  for t in 64x32x32 32x64x32 32x32x32 64x64x32; do
    echo -n "LN in the epilogue, tile $t     : "; AVSEP_LNX=1 AVSEP_LNX_TILE=$t one --steps 200 --rounds 5
  done
  echo -n "LN in the epilogue, picked tiles with BK = 32: "; AVSEP_LNX=1 AVSEP_LNX_BK32=1 one --steps 200 --rounds 5
done > $O/lnx_tiles.txt 2>&1
cat $O/lnx_tiles.txt
echo done
