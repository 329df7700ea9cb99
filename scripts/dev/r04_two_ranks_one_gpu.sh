#!/bin/bash
# C4 with two ranks (two processes) sharing the ONE GPU of the lease: the in-kernel merge through hipIpc-mapped
# buffers (--merge peers) against the RCCL-style scheme (--merge avg, gloo here: RCCL wants a device per rank) and
# against the single process.  Output: gpurun_out/two_ranks.txt
export SGDNET_BENCH_BACKEND=gloo SGDNET_BENCH_ONE_GPU=1
out=gpurun_out/two_ranks.txt
: > $out
line() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c=d.get('convergence') or {}
print(sys.argv[2], '| epochs/s', round(d['value'],1), '| ms per epoch', round(d['ms_per_step'],4), '| shards per rank', d['config'].get('virtual_shards'), '| window', d['config']['batch'], '| epochs to 1e-6', c.get('epochs'), '| deviance', c.get('deviance'), '| merge:', d['config']['merge'][:110])
" "$1" "$2" >> $out; }
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/tr_1.json 2>/dev/null && line gpurun_out/tr_1.json "1 process, 8 shards, 248 workgroups       "
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --vshards 4 > gpurun_out/tr_1v4.json 2>/dev/null && line gpurun_out/tr_1v4.json "1 process, 4 shards                        "
timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --merge peers > gpurun_out/tr_2p.json 2>gpurun_out/tr_2p.err && line gpurun_out/tr_2p.json "2 processes x 4 shards, in-kernel merge    "
timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --merge avg > gpurun_out/tr_2a.json 2>gpurun_out/tr_2a.err && line gpurun_out/tr_2a.json "2 processes x 4 shards, all-reduce of deltas"
cat $out
