#!/bin/bash
# round 3: tile of the small plain GEMMs (out-projection, FFN-2) inside the step (developer override AVSEP_SMALL_TILE)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03af; mkdir -p $O; cd $R
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing'].get('ms_per_step_min'), d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
export AVSEP_LIB=dev
for i in 1 2; do
  echo -n "picked (32x32x64)        : "; one --steps 200 --rounds 5
  for t in 32x64x32 32x64x64 64x32x32 64x32x64 32x32x32; do echo -n "AVSEP_SMALL_TILE=$t : "; AVSEP_SMALL_TILE=$t one --steps 200 --rounds 5; done
done > $O/small_tile.txt 2>&1
cat $O/small_tile.txt
