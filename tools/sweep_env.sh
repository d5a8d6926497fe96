run() { env "$@" python bench.py --no-cpu --no-profile --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
run AVSEP_TLN64=100000
run X=1
run AVSEP_TLN64=300
run AVSEP_TLN64=700
run AVSEP_TLN64=100000
run X=1
run AVSEP_TLN64=100
