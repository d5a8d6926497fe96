#!/bin/bash
# round 3: upper bound for an "LN in the epilogue" GEMM: the step with every LN-fused GEMM replaced by the plain GEMM launch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03q; mkdir -p $O; cd $R
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing'].get('ms_per_step_min'), d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
export AVSEP_LIB=dev
for i in 1 2 3; do
  echo -n "shipped (LN inside the GEMM prologue): "; one --steps 200 --rounds 5
  echo -n "plain GEMM launches instead (timing) : "; AVSEP_LN_AS_PLAIN=1 one --steps 200 --rounds 5
done > $O/ln_as_plain.txt 2>&1
echo done
