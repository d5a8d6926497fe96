#!/bin/bash
# round 3: attention + out-projection + residual in one launch (short sequences) and the load-first short attention kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03n; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing'].get('ms_per_step_min'), d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
export AVSEP_LIB=dev
for i in 1 2 3; do
  echo -n "fused attention+proj : "; one --steps 200 --rounds 5
  echo -n "separate launches    : "; AVSEP_NO_ATTN_PROJ=1 one --steps 200 --rounds 5
done > $O/attn_proj_ab.txt 2>&1
unset AVSEP_LIB
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2>$O/driver_cmd.err; echo "driver cmd done"
python3 tools/attn_bench.py 2>/dev/null | grep -v Warn | head -30 > $O/attn_bench.txt
