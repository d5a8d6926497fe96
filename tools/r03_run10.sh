#!/bin/bash
# round 3: does the audio branch progress beside the conv stack when its workgroups FIT there (16 KB of LDS each)?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03l; mkdir -p $O; cd $R
export AVSEP_LIB=dev AVSEP_SCHEDULE=fork
python3 tools/stamps.py cfg2 > $O/stamps_base.txt 2>&1
AVSEP_GEMM_TILE=32x32x32 AVSEP_NO_LN_FUSE=1 python3 tools/stamps.py cfg2 > $O/stamps_small_lds.txt 2>&1
AVSEP_GEMM_TILE=32x32x32 AVSEP_NO_LN_FUSE=1 AVSEP_CONV_GRID=200 python3 tools/stamps.py cfg2 > $O/stamps_small_lds_grid200.txt 2>&1
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing'].get('ms_per_step_min'), d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for i in 1 2; do
  echo -n "base                 : "; one --steps 200 --rounds 5
  echo -n "32x32x32 + no LN fuse: "; AVSEP_GEMM_TILE=32x32x32 AVSEP_NO_LN_FUSE=1 one --steps 200 --rounds 5
done > $O/small_lds_ab.txt 2>&1
echo done
