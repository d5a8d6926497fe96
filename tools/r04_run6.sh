#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O; cd $R
timeout -k 5 70 python3 tools/chain_debug.py 5 8 0 2 > $O/dbg5.txt 2>&1; tail -3 $O/dbg5.txt
AVSEP_LIB=dev AVSEP_CHAIN_DBG=1 timeout -k 5 70 python3 tools/chain_debug.py 32 8 0 2 2>&1 | grep -v "amdgpu.ids\|^enqueued\|forward 0\|forward 1" | tail -27 > $O/phase_stamps_xl.txt; cat $O/phase_stamps_xl.txt
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chained" > $O/chain_test.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/chain_test.txt
one() { timeout -k 10 100 python3 bench.py --no-cpu --no-profile --steps 200 --warmup 20 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
export AVSEP_LIB=dev
grep -q passed $O/chain_test.txt && for i in 1; do
  echo -n "schedule 0                : "; one
  for wp in 4 3 2; do for sk in "8 0" "1 0.5" "2 2.5"; do set -- $sk
    echo -n "schedule 2 g$1 s$2 wgpc $wp : "; AVSEP_CHAIN_WGPC=$wp one --schedule 2 --chain-group $1 --chain-skew $2
  done; done
done 2>&1 | tee $O/ab_chain_xl.txt
