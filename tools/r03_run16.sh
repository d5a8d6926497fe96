#!/bin/bash
# round 3: LayerNorm-in-the-epilogue GEMM (AMODE_LNX): op test, golden parity with the switch on, same-run A/B of the step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; cd $R
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "epilogue_form or rejects_bad_forms" > $O/op_test.txt 2>&1; echo "op test rc=$?" 
tail -5 $O/op_test.txt
AVSEP_LIB=dev AVSEP_LNX=1 timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "goldens or full_batch or edge_shapes" > $O/golden_lnx.txt 2>&1; echo "golden rc=$?"
tail -5 $O/golden_lnx.txt
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing'].get('ms_per_step_min'), d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
export AVSEP_LIB=dev
for i in 1 2 3; do
  echo -n "shipped (LN inside the GEMM prologue): "; one --steps 200 --rounds 5
  echo -n "LN in the epilogue                   : "; AVSEP_LNX=1 one --steps 200 --rounds 5
done > $O/lnx_ab.txt 2>&1
cat $O/lnx_ab.txt
AVSEP_LNX=1 python3 bench.py --no-cpu --steps 100 --rounds 3 > $O/bench_lnx_profile.json 2> $O/bench_lnx_profile.err
python3 bench.py --no-cpu --steps 100 --rounds 3 > $O/bench_shipped_profile.json 2>> $O/bench_lnx_profile.err
echo done
