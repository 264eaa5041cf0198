#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python tools/trace_invert.py 8 2>&1 | grep "invert" | tail -3 | tr '\n' ' '; echo; }
for i in 1 2; do
run CURV_SHARED_SIDE=0
run CURV_SHARED_SIDE=1
run CURV_SHARED_SIDE=1 CURV_SMALL_START=15
run CURV_SHARED_SIDE=1 CURV_NEAR_SIDE=1
done
