#!/bin/bash
# tools/dryrun_ranks.sh N [bench args] -- FUNCTIONAL run of bench.py --gpus N as the driver launches it, with the N ranks sharing the
# one GPU of the box and a gloo rendezvous (RBS_BENCH_BACKEND=gloo: never a performance number): the line lands in gpurun_out/.
N=${1:-2}; shift
out=gpurun_out/dryrun_$N; mkdir -p $out
( time RBS_BENCH_BACKEND=gloo timeout 2400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
    --master-port $((29400+N)) bench.py --gpus $N --steps 5 --warmup 2 "$@" ) > $out/bench_$N.json 2> $out/bench_$N.err
grep -c '^{' $out/bench_$N.json; tail -3 $out/bench_$N.err
