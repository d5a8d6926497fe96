#!/bin/bash
# round 3: audio front-end (+ first layers) with <= 24 KB workgroups so that they fit beside the conv stack
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03m; mkdir -p $O; cd $R
export AVSEP_LIB=dev
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing'].get('ms_per_step_min'), d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for i in 1 2 3; do
  for n in 0 1 2 3; do echo -n "audio small $n: "; AVSEP_AUDIO_SMALL=$n one --steps 200 --rounds 5; done
done > $O/audio_small_ab.txt 2>&1
AVSEP_SCHEDULE=fork AVSEP_AUDIO_SMALL=2 python3 tools/stamps.py cfg2 > $O/stamps_small2.txt 2>&1
AVSEP_SCHEDULE=fork AVSEP_AUDIO_SMALL=1 python3 tools/stamps.py cfg2 > $O/stamps_small1.txt 2>&1
echo done
