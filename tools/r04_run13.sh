#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04m; mkdir -p $O; cd $R
timeout -k 10 100 python3 tools/gemm_split_debug.py 2>&1 | grep -v amdgpu.ids | tail -5
timeout -k 10 200 python3 tools/gemm_split_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_split_probe.txt
