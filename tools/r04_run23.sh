#!/bin/bash
# round 4, run 23: the 256x128 split-precision kernel -- precision test, then the shape probe (both variants against the fp32 GEMM)
set -e -o pipefail
mkdir -p gpurun_out
export AVSEP_LIB=dev
timeout -k 10 240 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split" > gpurun_out/run23_test.txt 2>&1 || { tail -30 gpurun_out/run23_test.txt; exit 1; }
tail -3 gpurun_out/run23_test.txt
AVSEP_SPLIT_VARIANT=2 timeout -k 10 240 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split" > gpurun_out/run23_test_v2.txt 2>&1 || { tail -30 gpurun_out/run23_test_v2.txt; exit 1; }
tail -3 gpurun_out/run23_test_v2.txt
timeout -k 10 300 python3 tools/gemm_split_probe.py > gpurun_out/run23_probe.txt 2>&1 || { tail -30 gpurun_out/run23_probe.txt; exit 1; }
cat gpurun_out/run23_probe.txt
