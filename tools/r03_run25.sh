#!/bin/bash
# round 3: audio convolutions (TAPS3) read a zero row for out-of-sequence taps instead of predicate + select: parity + A/B vs HEAD's library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03ad; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_reference_suite_gpu.py -q -m gpu > $O/parity.txt 2>&1; echo "parity rc=$?"; tail -3 $O/parity.txt
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing'].get('ms_per_step_min'), d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for i in 1 2 3; do
  echo -n "previous library: "; AVSEP_LIB=$R/av-separation-transformer_amd/lib/libavsep_hip_prev.so one --steps 200 --rounds 5
  echo -n "zero-row taps   : "; one --steps 200 --rounds 5
done > $O/ab.txt 2>&1
cat $O/ab.txt
python3 bench.py --no-cpu --steps 100 --rounds 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in d['kernels']: print('%-44s x%4.1f %7.2f us %6.1f TF'%(k['name'][:44],k['calls_per_step'],k['avg_us'],k['tflops']))" > $O/kernels.txt
cat $O/kernels.txt
