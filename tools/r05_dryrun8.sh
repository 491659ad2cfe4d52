#!/bin/bash
# functional dry run of the driver's 8-rank launch on ONE GPU (gloo rendezvous, ranks share the device): never a performance number
out=gpurun_out/r05_dry8; mkdir -p $out
for N in 8 4; do
( time RBS_BENCH_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29400+N)) bench.py --gpus $N --steps 5 --warmup 2 --particles 500 --no-configs-leg ) > $out/bench_$N.json 2> $out/bench_$N.err
echo "N=$N rc=$?"
python - <<PY
import json
l=[x for x in open("$out/bench_$N.json") if x.startswith("{")]
if not l: print("NO LINE"); raise SystemExit
d=json.loads(l[-1])
for k in ("n_gpus","value","ms_per_step","multi_gpu_equals_single","ipc_attach_ok","peer_read_ok","rccl_ranks_seen","multi_gpu_check_remote_children","multi_gpu_check_max_abs_loglik_diff","multi_gpu_check_diagnosis","peer_step","remote_parent_frac","in_handle_equals_single","in_handle_rccl_ok","in_handle_max_abs_loglik_diff","tracker_fps_sharded_2000","tracker_fps_sharded_note"):
    print(k, d.get(k))
PY
tail -3 $out/bench_$N.err
done
