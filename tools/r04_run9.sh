#!/bin/bash
# round 4: extended shape fuzz of the drop-in (inference vs the numpy oracle, training vs the CPU port) on the final library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; mkdir -p $O; cd $R
AVSEP_FUZZ_INFER=500 AVSEP_FUZZ_TRAIN=80 timeout -k 10 1000 python3 -m pytest tests/test_fuzz_gpu.py -m gpu -q -x > $O/fuzz_extended.txt 2>&1; echo "rc=$?"; tail -4 $O/fuzz_extended.txt
