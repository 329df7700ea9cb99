#!/bin/bash
# run-time switches on ONE box.  usage: scripts/ab_envs.sh <tag> <workload> <shards> VAR=a VAR=b ...
set -eo pipefail
TAG="$1"; WL="$2"; VS="$3"; shift 3
mkdir -p gpurun_out/$TAG
for KV in "$@"; do
  echo "== $KV" | tee -a gpurun_out/$TAG/all.log
  env "$KV" timeout -k 10 300 python scripts/vshard_bench.py $WL $VS 2>&1 | tee -a gpurun_out/$TAG/all.log
done
