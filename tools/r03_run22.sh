#!/bin/bash
# round 3: long-sequence attention, wave priority around the MFMA clusters / the softmax (developer instances)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03z; mkdir -p $O; cd $R
export AVSEP_LIB=dev
for p in none 1 2 none 1 2; do
  echo "== AVSEP_ATTN_PRIO=$p"
  if [ $p = none ]; then python3 tools/attn_bench.py 2>/dev/null | grep "cfg3\|cfg5"; else AVSEP_ATTN_PRIO=$p python3 tools/attn_bench.py 2>/dev/null | grep "cfg3\|cfg5"; fi
done > $O/attn_prio.txt 2>&1
cat $O/attn_prio.txt
