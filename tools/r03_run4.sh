#!/bin/bash
# round 3: LDS-DMA GEMM A/B (shapes + cfg3/cfg5/train end to end), bit-identity tests, driver command on the product library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "tiles_bit_identical or op_linear" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2>$O/driver_cmd.err; echo "driver cmd done"
python3 tools/r03_dma_ab.py > $O/dma_shapes.txt 2>&1; echo "dma shapes done"
one() { python3 bench.py --no-cpu --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing'].get('ms_per_step_min'), d.get('one_step_at_a_time',{}).get('ms_per_step'))"; }
export AVSEP_LIB=dev
for i in 1 2; do
  for w in cfg3 cfg5; do
    echo -n "$w ring: "; one --workload $w --steps 20 --warmup 3 --rounds 3
    echo -n "$w dma : "; AVSEP_GEMM_DMA=1 one --workload $w --steps 20 --warmup 3 --rounds 3
  done
  echo -n "train ring: "; one --mode train --steps 10 --warmup 3 --rounds 3
  echo -n "train dma : "; AVSEP_GEMM_DMA=1 one --mode train --steps 10 --warmup 3 --rounds 3
done > $O/dma_end_to_end.txt 2>&1
echo "dma e2e done"
