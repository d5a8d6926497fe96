#!/bin/bash
# round 3: mask head with a straight-line epilogue: bit-identity test, same-run A/B; three steps in flight on the new kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03w; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu > $O/parity.txt 2>&1; echo "parity rc=$?"
tail -4 $O/parity.txt
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing'].get('ms_per_step_min'), d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
export AVSEP_LIB=dev
for i in 1 2 3; do
  echo -n "mask head, block-by-block epilogue: "; AVSEP_MASK_GENERAL=1 one --steps 200 --rounds 5
  echo -n "mask head, straight-line epilogue : "; one --steps 200 --rounds 5
done > $O/mask_ab.txt 2>&1
cat $O/mask_ab.txt
for n in 2 3 4; do echo -n "steps in flight $n: "; one --inflight $n --steps 200 --rounds 5; done > $O/inflight.txt 2>&1
cat $O/inflight.txt
python3 bench.py --no-cpu --steps 100 --rounds 3 > $O/bench_profile.json 2> $O/bench_profile.err
echo done
