#!/bin/bash
# round 3: in-launch slice merge of the weight-gradient kernel: parity + race screen, training step A/B, launch census
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python3 - > $O/train_ab.txt 2>&1 <<'PY'
import subprocess, sys, json
def run(merged):
    code = ("import sys; sys.argv=['bench.py','--mode','train','--steps','10','--warmup','3','--no-cpu']; sys.path.insert(0,'av-separation-transformer_amd'); "
            "from av_separation import _train as tr; tr.MERGED_WGRAD=%s; import runpy; runpy.run_path('bench.py', run_name='__main__')" % merged)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    d = json.loads(line[-1]); return d["value"], d["ms_per_step"]
for i in range(3):
    print("merged wgrad on ", *run(True)); print("merged wgrad off", *run(False))
PY
echo "train ab done"
timeout -k 10 300 python3 tools/train_trace.py > $O/train_trace.txt 2>&1; echo "trace done"
