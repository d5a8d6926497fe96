#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04s; mkdir -p $O; cd $R
for st in 0 4 8 16 32; do echo "== stagger $st"; AVSEP_SPLIT_STAGGER=$st timeout -k 10 100 python3 tools/gemm_split_probe.py 2>&1 | grep "16064\|4016" ; done | tee $O/split_stagger.txt
