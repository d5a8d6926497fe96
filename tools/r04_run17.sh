#!/bin/bash
# PMC passes over the split-precision GEMM (developer library routes avsep_op_linear to it with AVSEP_GEMM_SPLIT=1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04q; mkdir -p $O; cd $R
export AVSEP_LIB=dev AVSEP_GEMM_SPLIT=1
bash tools/pmc_gemm.sh "16064 2048 512" "16064 512 2048" > $O/pmc_gemm_split.txt 2>&1; cat $O/pmc_gemm_split.txt
