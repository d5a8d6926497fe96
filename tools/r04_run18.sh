#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04r; mkdir -p $O; cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "split_precision" 2>&1 | tail -3
timeout -k 10 200 python3 tools/gemm_split_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_split_probe.txt
export AVSEP_LIB=dev AVSEP_GEMM_SPLIT=1
bash tools/pmc_gemm.sh "16064 2048 512" > $O/pmc_gemm_split.txt 2>&1; tail -12 $O/pmc_gemm_split.txt
