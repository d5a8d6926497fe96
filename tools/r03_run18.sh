#!/bin/bash
# round 3: LayerNorm-in-the-epilogue GEMM as the default: parity suite, same-run A/B on cfg2, and the large configs with it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03u; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu > $O/parity.txt 2>&1; echo "parity rc=$?"
tail -8 $O/parity.txt
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing'].get('ms_per_step_min'), d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
export AVSEP_LIB=dev
for i in 1 2 3; do
  echo -n "LN inside the GEMM prologue (round 2)  : "; AVSEP_NO_LNX=1 one --steps 200 --rounds 5
  echo -n "LN in the epilogue (default)           : "; one --steps 200 --rounds 5
  echo -n "LN in the epilogue, 32x32x32 visual    : "; AVSEP_LNX_SMALL=1 one --steps 200 --rounds 5
done > $O/lnx_ab.txt 2>&1
cat $O/lnx_ab.txt
for wl in cfg3 cfg5; do
  for i in 1 2; do
    echo -n "$wl LayerNorm launch + GEMM (default): "; one --workload $wl --steps 20 --rounds 3
    echo -n "$wl LN in the epilogue (AVSEP_LNX=all): "; AVSEP_LNX=all one --workload $wl --steps 20 --rounds 3
  done
done > $O/lnx_big.txt 2>&1
cat $O/lnx_big.txt
echo done
