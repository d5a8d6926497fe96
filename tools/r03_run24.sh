#!/bin/bash
# round 3: 32-bit dropout hash: training tests + training bench (previous figure on this box class: 989 clips/s, 16.18 ms)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03ac; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_train_ops_gpu.py tests/test_train_gpu.py tests/test_fuzz_gpu.py -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
for i in 1 2; do
python3 bench.py --mode train --steps 10 --warmup 3 --no-cpu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', d['value'], d['ms_per_step'], d['timing']['ms_per_step_rounds'])"
done > $O/train.txt 2>&1; cat $O/train.txt
python3 - <<'PY' > $O/dropout_stats.txt 2>&1
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "av-separation-transformer_amd"))
import torch
from av_separation import _native
lib = _native.load(); dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
n = 1 << 24
x = torch.ones(n, device=dev); y = torch.empty_like(x)
for p in (0.1, 0.25, 0.5):
    for seed in (1, 1234, 0x9E3779B97F4A7C15, 2**63 + 12345):
        _native.check(lib.avsep_op_dropout(x.data_ptr(), y.data_ptr(), n, p, seed, st))
        k = (y > 0).float()
        kept = float(k.mean())
        # serial correlation of neighbours and of elements 2^k apart, and agreement with the mask of seed + 1
        cors = [float(((k[:-d] - kept) * (k[d:] - kept)).mean() / (kept * (1 - kept))) for d in (1, 2, 3, 16, 256, 4096, 65536)]
        _native.check(lib.avsep_op_dropout(x.data_ptr(), y.data_ptr(), n, p, seed + 1, st))
        k2 = (y > 0).float()
        cross = float(((k - kept) * (k2 - kept)).mean() / (kept * (1 - kept)))
        print(f"p={p} seed={seed:#x}: kept {kept:.5f} (expect {1-p:.5f}, sigma {((p*(1-p))/n)**0.5:.5f}); lag correlations " + " ".join(f"{c:+.4f}" for c in cors) + f"; vs seed+1 {cross:+.4f}")
PY
cat $O/dropout_stats.txt
echo done
