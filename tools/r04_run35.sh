#!/bin/bash
# round 4, run 35: the whole GPU test suite on the current build
R=$GRAFT_REPO_ROOT; cd $R
set -o pipefail
timeout -k 10 1100 python3 -m pytest tests/ -x -q -m gpu 2>&1 | tail -8
