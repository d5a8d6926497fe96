#!/bin/bash
# round 3: does the audio branch gain from room beside the conv stack?  (fork schedule, developer switches)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_min'], d['timing']['ms_per_step_max'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
export AVSEP_LIB=dev
for i in 1 2; do
  echo -n "base          : "; one --steps 200 --rounds 5
  echo -n "conv G1       : "; AVSEP_CONV_G1=1 one --steps 200 --rounds 5
  echo -n "conv G1 wgpc2 : "; AVSEP_CONV_G1=1 AVSEP_CONV_WGPC=2 one --steps 200 --rounds 5
  echo -n "conv grid 200 : "; AVSEP_CONV_GRID=200 one --steps 200 --rounds 5
  echo -n "conv nw4      : "; AVSEP_CONV_NW4=1 one --steps 200 --rounds 5
done > $O/conv_ab.txt 2>&1
echo "conv ab done"
AVSEP_SCHEDULE=fork python3 tools/stamps.py cfg2 > $O/stamps_fork.txt 2>&1
AVSEP_SCHEDULE=fork AVSEP_CONV_G1=1 python3 tools/stamps.py cfg2 > $O/stamps_fork_g1.txt 2>&1
AVSEP_CONV_DBG=1 python3 tools/one_fwd.py cfg2 2 2>&1 | grep "conv dbg" > $O/conv_dbg_g2.txt
AVSEP_CONV_DBG=1 AVSEP_CONV_G1=1 python3 tools/one_fwd.py cfg2 2 2>&1 | grep "conv dbg" > $O/conv_dbg_g1.txt
echo "stamps done"
