#!/bin/bash
# round 3: training-step launch removal (epilogue dropout, residual gradient inside LayerNorm backward), trace of what is left
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for i in 1 2; do python3 bench.py --mode train --steps 10 --warmup 3 --no-cpu 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done > $O/train.txt 2>&1
python3 - > $O/train_unfused.txt 2>&1 <<'PY'
import subprocess, sys, os
code = "import sys; sys.argv=['bench.py','--mode','train','--steps','10','--warmup','3','--no-cpu']; sys.path.insert(0,'av-separation-transformer_amd'); from av_separation import _train as tr; tr.EPILOGUE_DROPOUT=False; tr.FUSED_RESIDUAL_NORM=False; import runpy; runpy.run_path('bench.py', run_name='__main__')"
for i in range(2):
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    d = json.loads(line[-1]); print("train unfused", d["value"], d["ms_per_step"])
PY
echo "train done"
timeout -k 10 300 python3 tools/train_trace.py > $O/train_trace.txt 2>&1; echo "trace done"
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing'].get('ms_per_step_min'), d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for i in 1 2; do for w in cfg3 cfg5; do echo -n "$w: "; one --workload $w --steps 20 --warmup 3 --rounds 3; done; done > $O/big.txt 2>&1
export AVSEP_LIB=dev
for i in 1 2 3; do
  echo -n "inflight2 tail split   : "; one --steps 200 --rounds 5
  echo -n "inflight2 no tail split: "; AVSEP_TAIL_SPLIT=0 one --steps 200 --rounds 5
done > $O/tail_split_inflight.txt 2>&1
echo "all done"
