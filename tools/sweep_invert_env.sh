#!/bin/bash
# invert() of the ResNet-50 factors under a few settings of the sweep's environment knobs (same box, back to back)
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python tools/trace_invert.py 5 2>&1 | grep "invert" | tail -3 | tr '\n' ' '; echo; }
run CURV_NEAR_SIDE=0 CURV_INV_STREAM=0
run CURV_NEAR_SIDE=1 CURV_INV_STREAM=0
run CURV_NEAR_SIDE=0 CURV_INV_STREAM=2
run CURV_NEAR_SIDE=1 CURV_INV_STREAM=2
run CURV_NEAR_SIDE=1 CURV_INV_STREAM=1
run CURV_NEAR_SIDE=0 CURV_INV_STREAM=0
