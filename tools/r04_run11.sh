#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04k; mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/gemm_tail_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_tail_probe.txt
