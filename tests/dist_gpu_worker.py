"""Spawned by tests/test_gpu_parity.py::test_two_rank_sharding_on_one_gpu.
argv: port [slab_px]: the ranks' handles store their planes in slabs of slab_px floats (0 = whole planes) -- small enough
and every rank enlarges its slabs on its own schedule, so a migrating window may meet a receiver whose slabs are still
smaller (rbs_import_window grows them: ADVICE r4)."""
import os
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import scenarios as sc  # noqa: E402
from dbot_ros_amd import RbSensor, synth  # noqa: E402
from dbot_ros_amd import dist as rdist  # noqa: E402
from dbot_ros_amd import filter as flt  # noqa: E402

N, FRAMES = 24, 4


def inputs():
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=2 * N)
    with RbSensor(om, cam, P, max_particles=1) as s:
        rng = np.random.default_rng(0)
        frames = []
        for k in range(FRAMES):
            t = synth.truth_pose(1, frame=k)
            frames.append((t, synth.make_frame(s.render_depth(t), 120, 160, rng)))
    rng = np.random.default_rng(12)
    poses = [synth.particle_poses(t, N, rng, scale=2.0) for t, _ in frames]
    uniforms = [rng.random(N) for _ in frames]
    return om, cam, P, frames, poses, uniforms


def single():
    om, cam, P, frames, poses, uniforms = inputs()
    out = []
    with RbSensor(om, cam, P, max_particles=N) as s:
        s.reset()
        idx = np.zeros(N, np.int32)
        for k, (_, frame) in enumerate(frames):
            s.set_observation(frame)
            ll = s.loglikes_poses(poses[k], idx, update=True)
            parents = flt.multinomial_resample(flt.normalized_weights(ll), uniforms[k])
            out.append((ll.copy(), parents.copy()))
            idx = parents.astype(np.int32).copy()
    return out


def worker(rank, world, port, q, slab_px=0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        om, cam, P, frames, poses, uniforms = inputs()
        bounds = rdist.shard_bounds(N, world)
        with RbSensor(om, cam, P, max_particles=2 * int(np.diff(bounds).max()), slab_px=slab_px) as s:
            ss = rdist.ShardedSensor(s, N)
            ss.reset()
            res, moves = [], 0
            for k, (_, frame) in enumerate(frames):
                ss.set_observation(frame)
                ll = ss.loglikes(poses[k], update=True)
                parents = flt.multinomial_resample(flt.normalized_weights(ll), uniforms[k])
                res.append((ll.copy(), parents.copy()))
                moves += len(ss.resample(parents))
        q.put((rank, res, moves))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    port = int(sys.argv[1])
    slab_px = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ref = single()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q, slab_px)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][2] == got[1][2] and got[0][2] > 0, "no plane migrated"
    for _, res, _ in got:
        for (lr, pr), (l, p) in zip(ref, res):
            assert np.array_equal(l, lr), np.abs(l - lr).max()   # same device code, same planes: bitwise
            assert np.array_equal(p, pr)
    print("SHARDED_OK moves=%d" % got[0][2])
