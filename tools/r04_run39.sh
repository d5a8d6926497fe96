#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_shapes or split" -s 2>&1 | grep -v amdgpu.ids | tail -8
