#!/bin/bash
# same-box A/B of an environment switch of the library: bench.py's headline run with and without it, interleaved
#   gpurun -- 'bash tools/ab_env.sh 3 CURV_BIG_FAR=1000000000'
ROUNDS=$1; shift
for r in $(seq 1 $ROUNDS); do
  for setting in "" "$@"; do
    env $setting python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phases_ms']
print('[%s]' % '$setting', 'step %.2f update %.3f invert %.3f sample %.3f' % (d['ms_per_step'], p['update'], p['invert'], p['sample_and_replace']))"
  done
done
