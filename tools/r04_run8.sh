#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04h; mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/train_graph_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/train_graph_probe.txt
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chained or product_library" 2>&1 | tail -3
