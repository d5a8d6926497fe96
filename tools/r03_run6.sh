#!/bin/bash
# round 3: where does the LN-fused GEMM's time go, per tile shape?  (phase stamps of the developer build)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03h; mkdir -p $O; cd $R
export AVSEP_LIB=dev
for shp in "2016 768 256" "2016 1024 256" "1600 768 256" "1008 1024 256" "2016 512 256"; do
  for t in 32x32 32x64 64x32 64x64; do
    echo "== $shp tile $t"
    AVSEP_GEMM_DBG=all AVSEP_LN_TILE=$t python3 tools/gemm_ln_one.py $shp 4 2>&1 | grep "gemm dbg" | tail -2
  done
done > $O/ln_gemm_phases.txt 2>&1
python3 - > $O/devident.txt 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, ".")
import bench
print(bench.device_identity(torch.device("cuda:0")))
PY
echo done
