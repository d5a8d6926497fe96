#!/bin/bash
# round 3, first run of the paired schedule: GPU tests, pair-launch probe, schedule A/B on one device
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O; cd $R
timeout -k 10 400 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2>$O/driver_cmd.err; echo "driver cmd done"
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_min'], d['timing']['ms_per_step_max'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
export AVSEP_LIB=dev
for i in 1 2; do
  for r in 1 2 3 4; do
    echo -n "paired inflight$r: "; one --steps 200 --rounds 5 --inflight $r
    echo -n "fork   inflight$r: "; AVSEP_SCHEDULE=fork one --steps 200 --rounds 5 --inflight $r
  done
  echo -n "paired serial inflight1: "; AVSEP_SERIAL=1 one --steps 200 --rounds 5 --inflight 1
  echo -n "paired serial inflight2: "; AVSEP_SERIAL=1 one --steps 200 --rounds 5 --inflight 2
done > $O/schedule_ab.txt 2>&1
echo "schedule ab done"
for w in cfg3 cfg5; do
  echo -n "$w paired: "; one --workload $w --steps 20 --warmup 3 --rounds 3
  echo -n "$w fork  : "; AVSEP_SCHEDULE=fork one --workload $w --steps 20 --warmup 3 --rounds 3
done > $O/schedule_ab_big.txt 2>&1
echo "big ab done"
python3 tools/stamps.py cfg2 > $O/stamps_paired.txt 2>&1
python3 tools/r03_group_probe.py > $O/group_probe.txt 2>&1
echo "probe done"
