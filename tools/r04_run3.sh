#!/bin/bash
# round 4, GPU call 3: first run of the chained encoder layers -- bit identity, then same-run A/B of the schedules
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chained" > $O/chain_test.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/chain_test.txt
one() { timeout -k 10 200 python3 bench.py --no-cpu --no-profile --steps 200 --warmup 20 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for i in 1 2; do
  echo -n "schedule 0          : "; one
  echo -n "schedule 1 g8 s0    : "; one --schedule 1 --chain-group 8 --chain-skew 0
  echo -n "schedule 1 g8 s1    : "; one --schedule 1 --chain-group 8 --chain-skew 1
  echo -n "schedule 1 g4 s2.5  : "; one --schedule 1 --chain-group 4 --chain-skew 2.5
  echo -n "schedule 1 g16 s2.5 : "; one --schedule 1 --chain-group 16 --chain-skew 2.5
  echo -n "schedule 1 g1 s0.25 : "; one --schedule 1 --chain-group 1 --chain-skew 0.25
done 2>&1 | tee $O/ab_chain.txt
