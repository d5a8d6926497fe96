#!/bin/bash
# round 4, run 51: the whole GPU suite on the final tree -> profiles/r04_validation.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O; cd $R
unset AVSEP_LIB
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/r04_validation.txt 2>&1; tail -2 $O/r04_validation.txt
python3 -c "
import sys; sys.path.insert(0,'av-separation-transformer_amd')
from av_separation import _native
print('build', _native.load().avsep_build_id().decode())"
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
