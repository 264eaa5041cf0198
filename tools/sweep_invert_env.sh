#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python tools/trace_invert.py 8 2>&1 | grep "invert" | tail -3 | tr '\n' ' '; echo; }
for i in 1 2; do
run CURV_SMALL_FREE_CUS=0
run CURV_SMALL_FREE_CUS=16
run CURV_SMALL_FREE_CUS=8
run CURV_SMALL_FREE_CUS=16 CURV_FREE_CUS=48
run CURV_SMALL_FREE_CUS=24 CURV_FREE_CUS=48
run CURV_SMALL_FREE_CUS=16 CURV_FUSED_STEP=1
done
