#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O; cd $R
timeout -k 5 70 python3 tools/chain_debug.py 2 > $O/dbg2.txt 2>&1 && tail -4 $O/dbg2.txt && \
timeout -k 5 70 python3 tools/chain_debug.py 32 > $O/dbg32.txt 2>&1 && tail -4 $O/dbg32.txt && \
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chained" > $O/chain_test.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/chain_test.txt
one() { timeout -k 10 100 python3 bench.py --no-cpu --no-profile --steps 200 --warmup 20 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
grep -q passed $O/chain_test.txt && for i in 1 2; do
  echo -n "schedule 0          : "; one
  echo -n "schedule 1 g8 s0    : "; one --schedule 1 --chain-group 8 --chain-skew 0
  echo -n "schedule 1 g8 s1    : "; one --schedule 1 --chain-group 8 --chain-skew 1
  echo -n "schedule 1 g4 s2.5  : "; one --schedule 1 --chain-group 4 --chain-skew 2.5
  echo -n "schedule 1 g16 s2.5 : "; one --schedule 1 --chain-group 16 --chain-skew 2.5
  echo -n "schedule 1 g1 s0.25 : "; one --schedule 1 --chain-group 1 --chain-skew 0.25
done 2>&1 | tee $O/ab_chain.txt
