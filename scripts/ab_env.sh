#!/bin/bash
# A/B of a run-time switch on ONE box.  usage: scripts/ab_env.sh VAR=value <tag> [workload] [shards]
set -eo pipefail
KV="$1"; TAG="$2"; WL="${3:-C4}"; VS="${4:-1,8}"
mkdir -p gpurun_out/$TAG
timeout -k 10 400 python scripts/vshard_bench.py $WL $VS > gpurun_out/$TAG/default.log 2>&1
env "$KV" timeout -k 10 400 python scripts/vshard_bench.py $WL $VS > gpurun_out/$TAG/switched.log 2>&1
echo "== default"; cat gpurun_out/$TAG/default.log; echo "== $KV"; cat gpurun_out/$TAG/switched.log
