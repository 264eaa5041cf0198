#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python tools/trace_invert.py 8 2>&1 | grep "invert" | tail -3 | tr '\n' ' '; echo; }
for i in 1 2; do
run CURV_NBO=4
run CURV_NBO=5
run CURV_NBO=6
run CURV_NBO=7
run CURV_NBO=6 CURV_SMALL_START=15
run CURV_NBO=6 CURV_SMALL_START=45
run CURV_NBO=6 CURV_WIDE_PROD=512
run CURV_NBO=6 CURV_WIDE_NEAR=0
done
