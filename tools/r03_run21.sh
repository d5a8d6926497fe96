#!/bin/bash
# round 3: long-sequence attention, P V of the previous tile issued inside the softmax block: bit-identity tests, alone, in cfg3 / cfg5
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03x; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_train_ops_gpu.py -q -m gpu -k "attention" > $O/attn_tests.txt 2>&1; echo "attention tests rc=$?"
tail -4 $O/attn_tests.txt
python3 tools/attn_bench.py > $O/attn_alone.txt 2>&1; cat $O/attn_alone.txt
for wl in cfg3 cfg5; do
  python3 bench.py --workload $wl --no-cpu --steps 20 --rounds 3 > $O/bench_$wl.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/bench_$wl.json').read().strip().splitlines()[-1])
print('$wl', d['value'], d['ms_per_step'])
for k in d['kernels']:
    if 'attention' in k['name']: print('   ', k['name'], k['calls_per_step'], k['avg_us'], k['tflops'])
"
done
echo done
