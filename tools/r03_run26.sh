#!/bin/bash
# round 3: tile of the two audio convolutions (TAPS3) inside the step (developer override AVSEP_TAPS_TILE)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03ae; mkdir -p $O; cd $R
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing'].get('ms_per_step_min'), d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
export AVSEP_LIB=dev
for i in 1 2; do
  echo -n "picked (32x32x32 / 32x32x64): "; one --steps 200 --rounds 5
  for t in 32x64x32 64x32x32 32x32x32 64x64x32; do echo -n "AVSEP_TAPS_TILE=$t       : "; AVSEP_TAPS_TILE=$t one --steps 200 --rounds 5; done
done > $O/taps_tile.txt 2>&1
cat $O/taps_tile.txt
