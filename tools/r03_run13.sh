#!/bin/bash
# round 3: N single-stream steps in flight (one hardware queue each) against the shipped 2 x two-stream steps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03o; mkdir -p $O; cd $R
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing'].get('ms_per_step_min'), d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
export AVSEP_LIB=dev
for i in 1 2; do
  echo -n "two-stream steps, 2 in flight : "; one --steps 200 --rounds 5
  for r in 2 3 4; do echo -n "single-stream steps, $r in flight: "; AVSEP_SERIAL=1 one --steps 200 --rounds 5 --inflight $r; done
  for r in 3 4; do echo -n "single-stream, no tail split... $r: "; AVSEP_SERIAL=1 AVSEP_TAIL_SPLIT=0 one --steps 200 --rounds 5 --inflight $r; done
done > $O/serial_inflight.txt 2>&1
echo done
