#!/bin/bash
# Epochs-to-tolerance of N ranks x V virtual shards (an N*V-way average) with the HIP kernels, N ranks
# sharing ONE GPU over gloo (the development pool hands out one GPU): statistical side of the
# multi-GPU design only -- the epochs/s of such a run mean nothing.
export SGDNET_BENCH_BACKEND=gloo SGDNET_BENCH_ONE_GPU=1
for cfg in "$@"; do
  set -- $cfg; N=$1; V=$2
  timeout -k 10 900 python3 bench.py $EXTRA --gpus $N --vshards $V --steps 2 --warmup 1 --no-cpu-baseline --conv-max-epochs ${CONV_MAX:-150} 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d.get('convergence',{})
print('N=$N V=$V ranks_seen', d.get('n_ranks_seen'), 'merge:', d['config']['merge'][:90], '| epochs to 1e-6:', c.get('epochs'), 'converged', c.get('converged'))"
done
