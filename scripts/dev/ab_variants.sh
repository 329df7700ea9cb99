#!/bin/bash
# Runs the same short bench on every library variant under build/variants/ (one GPU box).
#   scripts/dev/ab_variants.sh "<bench args>" name1 name2 ...
set -uo pipefail
bargs=$1; shift
mkdir -p gpurun_out/ab
for name in "$@"; do
  lib=$PWD/build/variants/libsgdnet_hip_$name.so
  [ "$name" == "default" ] && lib=$PWD/sgdnet_amd/lib/libsgdnet_hip.so
  SGDNET_LIB_PATH=$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-convergence --steps 5 --warmup 1 $bargs > gpurun_out/ab/$name.log 2>&1
  tail -1 gpurun_out/ab/$name.log | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('%-12s ms/epoch %.3f  gather %.1f us x %d  sweep %.1f us  frac %.4f' % ('$name', d['ms_per_step'], r['avg_launch_us'], r['launches'], r['sweep_avg_launch_us'], r['frac']))
except Exception as e:
    print('$name', 'failed', e)
"
done
