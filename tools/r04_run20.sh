#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04t; mkdir -p $O; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_valu_share.hip -o /tmp/mbv 2>/dev/null && timeout -k 5 120 /tmp/mbv | tee $O/mfma_bf16_valu_share.txt
