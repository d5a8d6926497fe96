#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O; cd $R
timeout -k 5 120 python3 tools/chain_layer_ab.py 32 200 2>&1 | grep -v amdgpu.ids | tee $O/ab_audio_encoder_alone.txt
for wp in 2 1; do echo "== AVSEP_CHAIN_WGPC=$wp"; AVSEP_LIB=dev AVSEP_CHAIN_WGPC=$wp timeout -k 5 120 python3 tools/chain_layer_ab.py 32 200 2>&1 | grep "round 2"; done | tee -a $O/ab_audio_encoder_alone.txt
AVSEP_LIB=dev AVSEP_CHAIN_DBG=1 timeout -k 5 120 python3 tools/chain_layer_ab.py 32 1 2>&1 | grep "chain dbg" | tail -26 > $O/phase_stamps_alone.txt; cat $O/phase_stamps_alone.txt
