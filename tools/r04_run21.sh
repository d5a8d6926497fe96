#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04u; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $O/validation.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/validation.txt
one() { timeout -k 10 200 python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for w in cfg3 cfg5; do echo -n "$w: "; one --workload $w --steps 20 --warmup 3 --rounds 5; done | tee $O/bench_split_conv_mask.txt
