"""Particle sharding across ranks, on CPU: world_size-2 gloo processes, each driving an ORACLE
instance for its shard through dbot_ros_amd.dist.ShardedSensor, must reproduce the
single-process run on the multiset of (parent, log-likelihood) -- incl. cross-rank plane
migration after resampling."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_binding as ob
import scenarios as sc
from dbot_ros_amd import dist as rdist
from dbot_ros_amd import filter as flt
from dbot_ros_amd import synth

N, FRAMES = 10, 3   # 10 particles over 2 ranks: shards of 5


def test_shard_bounds_and_placement():
    assert rdist.shard_bounds(10, 4).tolist() == [0, 3, 6, 8, 10]
    b = rdist.shard_bounds(8, 2)
    # all mass on rank 0's particle 1: 4 children stay, 4 migrate to rank 1
    parents = np.full(8, 1)
    rank, slot = rdist.place_children(parents, b)
    assert sorted(rank.tolist()) == [0, 0, 0, 0, 1, 1, 1, 1]
    for r in (0, 1):
        assert sorted(slot[rank == r].tolist()) == [0, 1, 2, 3]
    # identity parents: nobody moves
    rank, slot = rdist.place_children(np.arange(8), b)
    assert rank.tolist() == [0, 0, 0, 0, 1, 1, 1, 1] and slot.tolist() == [0, 1, 2, 3, 0, 1, 2, 3]


def _inputs():
    om, cam, P = sc.make_scene(("m1_l2",), 80, 60, max_particles=2 * N)
    o = ob.Oracle(om, cam, P, max_particles=1)
    frames = sc.make_frames(o, 1, FRAMES, seed=6)
    rng = np.random.default_rng(12)
    poses = [synth.particle_poses(t, N, rng, scale=2.0) for t, _ in frames]
    uniforms = [rng.random(N) for _ in frames]
    return om, cam, P, frames, poses, uniforms


def _single_process():
    om, cam, P, frames, poses, uniforms = _inputs()
    o = ob.Oracle(om, cam, P, max_particles=N, mode=ob.EAGER)
    o.reset()
    idx = np.zeros(N, np.int32)
    out = []
    for k, (_, frame) in enumerate(frames):
        o.set_observation(frame)
        ll = o.loglikes_poses(poses[k], idx, update=True)
        parents = flt.multinomial_resample(flt.normalized_weights(ll), uniforms[k])
        out.append((ll.copy(), parents.copy()))
        idx = parents.astype(np.int32).copy()
    return out


class _OracleWithImport(ob.Oracle):
    """The sharded driver needs set_occlusion (plane import); the oracle's C side has it."""

    def set_occlusion(self, slot, plane):
        import ctypes as C
        buf = np.ascontiguousarray(plane, dtype=np.float32)
        self._lib.orc_set_occlusion(self._h, int(slot), buf.ctypes.data_as(C.POINTER(C.c_float)))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        om, cam, P, frames, poses, uniforms = _inputs()
        bounds = rdist.shard_bounds(N, world)
        # device rule (EAGER): a plane is self-contained, so moving it between ranks is exact
        o = _OracleWithImport(om, cam, P, max_particles=2 * int(np.diff(bounds).max()), mode=ob.EAGER)
        ss = rdist.ShardedSensor(o, N)
        ss.reset()
        res, n_moves = [], 0
        for k, (_, frame) in enumerate(frames):
            ss.set_observation(frame)
            ll = ss.loglikes(poses[k], update=True)
            parents = flt.multinomial_resample(flt.normalized_weights(ll), uniforms[k])
            res.append((ll.copy(), parents.copy()))
            n_moves += len(ss.resample(parents))
        q.put((rank, res, n_moves))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_run_matches_single_process():
    ref = _single_process()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, moves = {}, {}
    for _ in procs:
        r, res, m = q.get(timeout=300)
        got[r], moves[r] = res, m
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert moves[0] == moves[1] and moves[0] > 0, "the scenario must exercise cross-rank plane migration"
    for rank in (0, 1):
        for (ll_ref, par_ref), (ll, par) in zip(ref, got[rank]):
            # same particles (poses are indexed by global child id), same inherited planes
            assert np.abs(ll - ll_ref).max() <= 1e-9 * max(1.0, np.abs(ll_ref).max())
            assert np.array_equal(par, par_ref)
