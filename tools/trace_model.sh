#!/bin/bash
# kernel timeline (lines LO..HI of the last call) of invert() on the ResNet-50 factors: tools/trace_model.sh LO HI TAG [finalizes per call]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
LO="${1:-0}"; HI="${2:-100}"; TAG="${3:-trm}"; PER="${4:-2}"
rm -rf gpurun_out/$TAG; mkdir -p gpurun_out/$TAG
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG/raw -- python tools/trace_invert.py 2 > gpurun_out/$TAG/log.txt 2>&1
python tools/trace_timeline.py gpurun_out/$TAG/raw $LO $HI $PER > gpurun_out/$TAG/timeline.txt
python tools/trace_buckets.py gpurun_out/$TAG/raw 500 $PER > gpurun_out/$TAG/buckets.txt
rm -rf gpurun_out/$TAG/raw
