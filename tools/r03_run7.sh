#!/bin/bash
# round 3: large-workgroup instances of the LN-fused GEMM: bit-identity, per-shape sweep, phase stamps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03i; mkdir -p $O; cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "ln_linear" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python3 tools/r03_ln_sweep.py > $O/ln_sweep.txt 2>&1; echo "sweep done"
export AVSEP_LIB=dev
for shp in "2016 768 256" "2016 1024 256"; do
  for t in 64x96x8 128x64x16 64x128x16 64x64x8; do
    echo "== $shp tile $t"
    AVSEP_GEMM_DBG=all AVSEP_LN_TILE=$t python3 tools/gemm_ln_one.py $shp 4 2>&1 | grep "gemm dbg" | tail -2
  done
done > $O/ln_gemm_phases_big.txt 2>&1
echo done
