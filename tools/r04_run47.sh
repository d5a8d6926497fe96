#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
export AVSEP_LIB=dev AVSEP_SPLIT_DBG=1 AVSEP_SPLIT_VARIANT=2
timeout -k 10 300 python3 tools/gemm_split_probe.py 2>&1 | grep "split dbg" | sort | uniq -c | sort -rn | head -30
