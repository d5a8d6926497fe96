#!/bin/bash
# round 4, GPU call 1: the full GPU suite on the pilot-shifted LayerNorm-epilogue build + same-run A/B against round 3's library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/validation.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/validation.txt
bash tools/ab_bench.sh --steps 200 --warmup 20 > $O/ab_lnx_pilot.txt 2>&1; cat $O/ab_lnx_pilot.txt
