#!/bin/bash
# round 3: graph-executable alternation A/B, new parity tests, baselines of the other workloads on this device
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cp gpurun_out/grad_gate_*.txt $O/ 2>/dev/null
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_min'], d['timing']['ms_per_step_max'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
export AVSEP_LIB=dev
for i in 1 2 3; do
  for r in 2 3; do
    echo -n "execs 2 inflight$r: "; one --steps 200 --rounds 5 --inflight $r
    echo -n "execs 1 inflight$r: "; AVSEP_GRAPH_EXECS=1 one --steps 200 --rounds 5 --inflight $r
  done
done > $O/graph_execs_ab.txt 2>&1
echo "execs ab done"
for w in cfg3 cfg5; do
  echo -n "$w execs 2: "; one --workload $w --steps 20 --warmup 3 --rounds 3
  echo -n "$w execs 1: "; AVSEP_GRAPH_EXECS=1 one --workload $w --steps 20 --warmup 3 --rounds 3
done > $O/graph_execs_ab_big.txt 2>&1
unset AVSEP_LIB
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2>$O/driver_cmd.err; echo "driver cmd done"
python3 bench.py --mode train --steps 10 --warmup 3 --no-cpu > $O/train_cfg4.json 2>/dev/null; echo "train done"
