#!/bin/bash
# invert() of the ResNet-50 factors under a few settings of the sweep's environment knobs (same box, back to back)
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python tools/trace_invert.py 5 2>&1 | grep "invert" | tail -3 | tr '\n' ' '; echo; }
run CURV_SMALL_MASKED=0
run CURV_SMALL_MASKED=1
run CURV_SMALL_MASKED=1 CURV_SMALL_START=0
run CURV_SMALL_MASKED=1 CURV_SMALL_START=15
run CURV_SMALL_MASKED=1 CURV_FREE_CUS=48
run CURV_SMALL_MASKED=1 CURV_FREE_CUS=16
run CURV_SMALL_MASKED=1 CURV_LATENCY_MAX=1000 CURV_ONE_GROUP=0
run CURV_SMALL_MASKED=0
