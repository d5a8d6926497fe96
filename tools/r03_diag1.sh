#!/bin/bash
# round-3 diagnostics of the cfg2 step on the current build (one device, one run)
export AVSEP_LIB=dev   # developer switches exist only in libavsep_hip_dev.so (make dev)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O; cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2>$O/driver_cmd.err
echo "driver cmd done"
python3 tools/stamps.py cfg2 > $O/stamps.txt 2>&1
echo "stamps done"
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for i in 1 2; do
  echo -n "inflight2 steps200: "; one --steps 200
  echo -n "inflight1 steps200: "; one --steps 200 --inflight 1
  echo -n "inflight1 SERIAL  : "; AVSEP_SERIAL=1 one --steps 200 --inflight 1
  echo -n "inflight2 SERIAL  : "; AVSEP_SERIAL=1 one --steps 200 --inflight 2
  echo -n "inflight1 notail  : "; AVSEP_TAIL_SPLIT=0 one --steps 200 --inflight 1
  echo -n "inflight1 eager   : "; one --steps 200 --inflight 1 --no-graph
done > $O/modes.txt 2>&1
echo "modes done"
