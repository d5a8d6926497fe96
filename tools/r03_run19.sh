#!/bin/bash
# round 3: full GPU suite on the LayerNorm-in-the-epilogue build + BK A/B of the small plain tiles inside the step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03v; mkdir -p $O; cd $R
timeout -k 10 1000 python3 -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"
tail -6 $O/gpu_tests.txt
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing'].get('ms_per_step_min'), d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
export AVSEP_LIB=dev
for i in 1 2 3; do
  echo -n "default                       : "; one --steps 200 --rounds 5
  echo -n "BK = 32 on the small plain tiles: "; AVSEP_BK32=plain one --steps 200 --rounds 5
  echo -n "BK = 32 on every small tile     : "; AVSEP_BK32=all one --steps 200 --rounds 5
done > $O/bk32_ab.txt 2>&1
cat $O/bk32_ab.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 1500 $O/bench_driver.json
echo done
