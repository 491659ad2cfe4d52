#!/usr/bin/env python3
"""Tracker FPS with the particles sharded over the GPUs of one node:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
         --master-port P tools/tracker_fps_dist.py [particles=20000] [frames=30]
Every rank runs the same host filter on the same seed (all particle states, 96 B each); it
evaluates its own shard on its GPU, the log-likelihoods are all-gathered (RCCL) and after each
resampling only the planes whose children landed on another rank migrate (device to device).
RBS_BENCH_BACKEND=gloo: functional run with ranks sharing GPUs (tests)."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dbot_ros_amd import CameraData, ObjectModel, RbSensor, RbSensorBuilder, pose, synth  # noqa: E402
from dbot_ros_amd.dist import ShardedRbSensor, shard_bounds  # noqa: E402
from dbot_ros_amd.tracker import ObjectTransitionBuilder, ParticleTracker, ParticleTrackerBuilder  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("RBS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    cols, rows = 640, 480
    v, f = synth.mesh_m1()
    om = ObjectModel([v], [f], center=True)
    cam = CameraData(synth.camera_matrix(cols, rows), rows, cols)
    shard = int(np.diff(shard_bounds(n, world)).max())
    P = RbSensorBuilder.Parameters(sample_count=2 * shard)
    sensor = RbSensor(om, cam, P, device_id=local, max_particles=2 * shard)   # own planes + staging
    rng = np.random.default_rng(0)
    frames = [synth.make_frame(sensor.render_depth(synth.truth_pose(1, frame=k)), rows, cols, rng, occluder=False)
              for k in range(n_frames + 1)]
    dev = torch.device("cuda", local) if backend == "nccl" else None
    ss = ShardedRbSensor(sensor, n, device=dev)
    trans = ObjectTransitionBuilder(ObjectTransitionBuilder.Parameters(part_count=1)).build()
    tr = ParticleTracker(trans, ss, om, ParticleTrackerBuilder.Parameters(evaluation_count=n), np.random.default_rng(1))
    Rt = synth.truth_pose(1, frame=0)[0]
    init = np.zeros(12)
    init[3:6] = pose.matrix_to_rotvec(Rt[:9].reshape(3, 3))
    init[0:3] = Rt[9:] - Rt[:9].reshape(3, 3) @ om.centers[0]
    tr.initialize([init])
    tr.track(frames[0])
    dist.barrier()
    t0 = time.perf_counter()
    for k in range(1, n_frames + 1):
        est = tr.track(frames[k])
    dist.barrier()
    dt = time.perf_counter() - t0
    Rt = synth.truth_pose(1, frame=n_frames)[0]
    err = float(np.linalg.norm(est[0:3] - (Rt[9:] - Rt[:9].reshape(3, 3) @ om.centers[0])))
    digest = float(np.sum(est))
    all_d = [None] * world
    dist.all_gather_object(all_d, digest)
    if rank == 0:
        assert all(d == all_d[0] for d in all_d), "ranks disagree on the estimate"
        print(json.dumps({"metric": "tracker FPS", "filter": "host, particles sharded", "evaluation_count": n,
                          "n_gpus": world, "value": n_frames / dt, "unit": "frames/s", "ms_per_frame": dt / n_frames * 1e3,
                          "planes_migrated": ss.moves, "resamplings": tr.n_resamplings, "final_position_error_m": err,
                          "estimate_digest": digest, "resolution": [cols, rows], "triangles": int(len(f))}), flush=True)
    ss.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
