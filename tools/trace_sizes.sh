#!/bin/bash
# kernel timeline of one chol_inv_lower call on a list of factor sizes: tools/trace_sizes.sh "4608" [first_line last_line]
# (run on the GPU box through gpurun; output under gpurun_out/trs/)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SIZES="$1"; LO="${2:-0}"; HI="${3:-100000}"; TAG="${4:-trs}"
cat > /tmp/one_sizes.py <<PY
import sys, torch
sys.path.insert(0, '.')
from curvature_amd import ops
dev = torch.device('cuda:0')
Fs = []
for i, n in enumerate([$SIZES]):
    torch.manual_seed(i)
    k = min(n + 8, 4096)
    X = torch.randn(n, k, device=dev)
    Fs.append((X @ X.t() / k).contiguous())
for _ in range(3):
    ops.chol_inv_lower(Fs, [1.0] * len(Fs), [1000.0] * len(Fs), check=False)
torch.cuda.synchronize()
PY
rm -rf gpurun_out/$TAG; mkdir -p gpurun_out/$TAG
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG/raw -- python /tmp/one_sizes.py > gpurun_out/$TAG/log.txt 2>&1
python tools/trace_timeline.py gpurun_out/$TAG/raw $LO $HI > gpurun_out/$TAG/timeline.txt
rm -rf gpurun_out/$TAG/raw
head -c 20000 gpurun_out/$TAG/timeline.txt
