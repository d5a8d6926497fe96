#!/bin/bash
# round 4: the whole GPU suite with the DEVELOPER library as the library of the process (its kernel instances went through the same
# gemm_tile.h / attn_tile.h refactor as the product's)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04l; mkdir -p $O; cd $R
AVSEP_LIB=dev timeout -k 10 1000 python3 -m pytest tests -m gpu -q --deselect tests/test_abi.py > $O/validation_dev_library.txt 2>&1; echo "rc=$?"; tail -6 $O/validation_dev_library.txt
