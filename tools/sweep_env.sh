run() { env "$@" python bench.py --no-cpu --no-profile --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
run X=1
run AVSEP_TLN=700
run AVSEP_TLN=300
run AVSEP_T6432=256
run AVSEP_T6432=256 AVSEP_T64=256
run AVSEP_TLN=700 AVSEP_T6432=256
run AVSEP_NO_LN_FUSE=1
run AVSEP_NO_SHORT_ATTN=1
run X=1
