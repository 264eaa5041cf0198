#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python tools/trace_invert.py 8 2>&1 | grep "invert" | tail -3 | tr '\n' ' '; echo; }
for i in 1 2; do
run CURV_FUSED_STEP=0
run CURV_FUSED_STEP=1
run CURV_FUSED_STEP=1 CURV_NBO=4
run CURV_FUSED_STEP=1 CURV_NBO=8
run CURV_FUSED_STEP=1 CURV_SMALL_START=15
done
