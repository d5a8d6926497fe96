#!/bin/bash
# round 4, run 27: split kernel rule (256x128 from 48 tiles) -- GPU parity suite of the forward, forwards in flight 1 / 2 / 3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04z; mkdir -p $O; cd $R
set -o pipefail
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4 || exit 1
one() { timeout -k 10 200 python3 bench.py --no-cpu --no-profile --no-also --no-quality "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
for w in cfg3 cfg5; do for f in 1 2 3 4; do
  echo -n "$w forwards in flight $f : "; one --workload $w --steps 20 --warmup 3 --rounds 5 --inflight $f
done; done 2>&1 | tee $O/ab_inflight_big_configs.txt
