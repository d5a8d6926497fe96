#!/bin/bash
# round 4, run 31: training tests (default path strict gates + the opt-in split-GEMM test), ABI / op tests
R=$GRAFT_REPO_ROOT; cd $R
set -o pipefail
timeout -k 10 900 python3 -m pytest tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -15 || exit 1
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split or linear" 2>&1 | tail -3
