#!/bin/bash
# round 3: conv_stack software pipeline (conv1 of the next pass inside conv3): parity, same-run A/B against the previous build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
B=$R/av-separation-transformer_amd/lib/libavsep_base.so
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing'].get('ms_per_step_min'), d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for i in 1 2 3; do
  echo -n "base: "; AVSEP_LIB=$B one --steps 200 --rounds 5
  echo -n "new : "; one --steps 200 --rounds 5
done > $O/conv_pipeline_ab.txt 2>&1
for i in 1 2; do
  for w in cfg3 cfg5; do echo -n "$w base: "; AVSEP_LIB=$B one --workload $w --steps 20 --warmup 3 --rounds 3; echo -n "$w new : "; one --workload $w --steps 20 --warmup 3 --rounds 3; done
done >> $O/conv_pipeline_ab.txt 2>&1
AVSEP_LIB=dev AVSEP_CONV_DBG=1 python3 tools/one_fwd.py cfg2 2 2>&1 | grep "conv dbg" > $O/conv_dbg.txt
python3 bench.py --no-cpu --steps 50 --rounds 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); [print(k) for k in d['kernels'][:3]]" > $O/kernels.txt
echo done
