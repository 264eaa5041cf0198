#!/bin/bash
# same-box A/B of library builds: bench.py's headline run with each library of the argument list, interleaved, ROUNDS times
#   gpurun -- 'bash tools/ab_libs.sh 3 curvature_amd/csrc/libcurv_hip.so tools/micro/libcurv_x.so'
ROUNDS=$1; shift
for r in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    CURV_ALT_LIB=$lib python3 tools/bench_with_lib.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phases_ms']
print('$lib', 'step %.2f update %.3f invert %.2f sample %.3f window %.3f frac %.3f' % (d['ms_per_step'], p['update'], p['invert'], p['sample_and_replace'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
  done
done
