#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04e; mkdir -p $O; cd $R
export AVSEP_LIB=dev
for bo in 2 48 127; do
  echo "== back-off $bo, op-major"; AVSEP_CHAIN_BACKOFF=$bo AVSEP_CHAIN_DBG=1 timeout -k 5 70 python3 tools/chain_debug.py 32 8 0 2>&1 | grep -v "amdgpu.ids\|^enqueued\|forward 0\|forward 1" | tail -32
done > $O/phase_stamps.txt 2>&1
cat $O/phase_stamps.txt
one() { timeout -k 10 100 python3 bench.py --no-cpu --no-profile --steps 200 --warmup 20 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for bo in 2 16 48 127; do for wp in 4 3 2 1; do
  echo -n "schedule 1 g8 s0 backoff $bo wgpc $wp : "; AVSEP_CHAIN_BACKOFF=$bo AVSEP_CHAIN_WGPC=$wp one --schedule 1
done; done 2>&1 | tee $O/ab_chain_backoff.txt
