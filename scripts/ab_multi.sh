#!/bin/bash
# several compile-time variants on ONE box (timing only).  usage: scripts/ab_multi.sh <tag> <workload> <shards> "<flags1>" "<flags2>" ...
set -eo pipefail
TAG="$1"; WL="$2"; VS="$3"; shift 3
mkdir -p gpurun_out/$TAG
for FL in "$@" ""; do
  EXTRA_FLAGS="$FL" ./build.sh > gpurun_out/$TAG/build.log 2>&1
  echo "== [$FL]" | tee -a gpurun_out/$TAG/all.log
  timeout -k 10 300 python scripts/vshard_bench.py $WL $VS 2>&1 | tee -a gpurun_out/$TAG/all.log
done
