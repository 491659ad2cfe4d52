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


# ------------------------------------------------- resamplings that no updating call follows
def _double_resample_script():
    """update, then per frame: two resamplings each followed by a READ-ONLY evaluation, then an
    updating evaluation under a third resampling -- the call pattern of a three-body tracker
    whose first two sampling blocks both resample."""
    om, cam, P, frames, poses, uniforms = _inputs()
    rng = np.random.default_rng(77)
    steps = []
    for k in range(len(frames)):
        steps.append([rng.integers(0, N, N) for _ in range(3)])
    return om, cam, P, frames, poses, steps


def _double_resample_single():
    om, cam, P, frames, poses, steps = _double_resample_script()
    o = ob.Oracle(om, cam, P, max_particles=N, mode=ob.EAGER)
    o.reset()
    idx = np.zeros(N, np.int32)
    out = []
    for k, (_, frame) in enumerate(frames):
        o.set_observation(frame)
        for j, parents in enumerate(steps[k]):
            idx = idx[parents].copy()
            out.append(o.loglikes_poses(poses[k], idx, update=(j == 2)).copy())
    return out


def _double_resample_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        om, cam, P, frames, poses, steps = _double_resample_script()
        shard = int(np.diff(rdist.shard_bounds(N, world)).max())
        o = _OracleWithImport(om, cam, P, max_particles=2 * shard, mode=ob.EAGER)
        ss = rdist.ShardedSensor(o, N)
        ss.reset()
        out, moves = [], 0
        for k, (_, frame) in enumerate(frames):
            ss.set_observation(frame)
            for j, parents in enumerate(steps[k]):
                moves += len(ss.resample(parents))
                out.append(ss.loglikes(poses[k], update=(j == 2)).copy())
        q.put((rank, out, moves))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_resampling_twice_without_update(world):
    ref = _double_resample_single()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 25500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_double_resample_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, out, moves in got:
        assert (moves > 0) == (world > 1)
        for a, b in zip(out, ref):
            assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(b).max()), np.abs(a - b).max()


# ---------------------------------------------------------------- the sharded TRACKER
def _tracker_run(sensor, om, frames, n, two_bodies=False):
    from dbot_ros_amd import pose
    from dbot_ros_amd.tracker import ObjectTransitionBuilder, ParticleTracker, ParticleTrackerBuilder
    parts = om.count_parts
    trans = ObjectTransitionBuilder(ObjectTransitionBuilder.Parameters(part_count=parts)).build()
    tr = ParticleTracker(trans, sensor, om, ParticleTrackerBuilder.Parameters(evaluation_count=n * parts),
                         np.random.default_rng(5))
    init = np.zeros(12 * parts)
    for b in range(parts):
        Rt = frames[0][0][b]
        init[12 * b + 3:12 * b + 6] = pose.matrix_to_rotvec(Rt[:9].reshape(3, 3))
        init[12 * b:12 * b + 3] = Rt[9:] - Rt[:9].reshape(3, 3) @ om.centers[b]
    tr.initialize([init])
    return np.array([tr.track(f) for _, f in frames]), tr.n_resamplings


def _tracker_inputs(parts):
    meshes = ("m1_l2", "box12", "m1_l2")[:parts]
    om, cam, P = sc.make_scene(meshes, 80, 60, max_particles=2 * N)
    o = ob.Oracle(om, cam, P, max_particles=1)
    return om, cam, P, sc.make_frames(o, parts, 5, seed=8)


def _tracker_worker(rank, world, port, parts, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        om, cam, P, frames = _tracker_inputs(parts)
        shard = int(np.diff(rdist.shard_bounds(N, world)).max())
        o = _OracleWithImport(om, cam, P, max_particles=2 * shard, mode=ob.EAGER)
        ss = rdist.ShardedRbSensor(o, N)
        ests, nres = _tracker_run(ss, om, frames, N)
        q.put((rank, ests, nres, ss.moves))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("parts", [1, 2, 3])
def test_two_rank_sharded_tracker_matches_single_process(parts):
    """ParticleTracker over a ShardedRbSensor on two gloo ranks (oracle evaluators) against the same
    tracker over one oracle: identical estimates -- with two bodies there are two sampling blocks
    per frame, i.e. read-only calls and resampling between them."""
    om, cam, P, frames = _tracker_inputs(parts)
    ref, nres_ref = _tracker_run(ob.Oracle(om, cam, P, max_particles=N, mode=ob.EAGER), om, frames, N)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 27500 + (os.getpid() % 2000) + parts
    procs = [ctx.Process(target=_tracker_worker, args=(r, 2, port, parts, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, ests, nres, moves in got:
        assert nres == nres_ref and nres > 0
        assert np.abs(ests - ref).max() <= 1e-9, np.abs(ests - ref).max()
    assert got[0][3] == got[1][3] and got[0][3] > 0, "no plane migrated"


# ---------------------------------------------------------------- the peer-read step (round 4)
PN, PSTEPS = 6, 5   # 6 particles per rank


def _peer_inputs(world):
    import torch
    n_all = PN * world
    om, cam, P = sc.make_scene(("m1_l2",), 80, 60, max_particles=2 * n_all)
    o = ob.Oracle(om, cam, P, max_particles=1)
    frames = sc.make_frames(o, 1, PSTEPS, seed=16)
    rng = np.random.default_rng(22)
    poses = [synth.particle_poses(t, n_all, rng, scale=2.0).reshape(n_all, -1) for t, _ in frames]   # by SLOT
    g = torch.Generator().manual_seed(5)
    uniforms = [torch.rand(n_all, dtype=torch.float64, generator=g) for _ in frames]
    return om, cam, P, frames, poses, uniforms


def _peer_single(world):
    import torch
    om, cam, P, frames, poses, uniforms = _peer_inputs(world)
    n_all = PN * world
    o = ob.Oracle(om, cam, P, max_particles=n_all, mode=ob.EAGER)
    o.reset()
    idx = np.zeros(n_all, np.int32)
    out = []
    for k, (_, frame) in enumerate(frames):
        o.set_observation(frame)
        ll = o.loglikes_poses(poses[k], idx, update=True)
        ps = rdist.global_resample(torch.from_numpy(ll), uniforms[k], temperature=30.0)
        out.append((ll.copy(), ps.numpy().copy()))
        idx = ps.numpy().astype(np.int32)
    return out


def _peer_worker(rank, world, port, q):
    import torch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        om, cam, P, frames, poses, uniforms = _peer_inputs(world)
        n, cap = PN, 2 * PN
        o = _OracleWithImport(om, cam, P, max_particles=cap, mode=ob.EAGER)
        o.reset()
        sent = [0]

        def evaluate(p, parent_idx, out):
            # on the CPU every remote parent was staged (min_share = 1): all parents are local slots
            local = (parent_idx.numpy() - rank * cap).astype(np.int32)
            assert ((local >= 0) & (local < cap)).all()
            out.copy_(torch.from_numpy(o.loglikes_poses(p.numpy(), local, update=True)))

        def stage(src, dst):
            # what rbs_stage_windows does by reading the owner's mapped planes, done here with messages:
            # every rank knows every rank's plan, the owner sends, the stager stores
            ops, landed, keep = [], [], []
            for r in range(world):
                _, s_r, d_r, _ = rdist.plan_shard(step.last_parents, n, cap, r, 1)
                for sg, dl in zip(s_r.tolist(), d_r.tolist()):
                    if dl < 0:
                        continue
                    owner, slot = divmod(sg, cap)
                    assert owner != r and slot < n
                    if owner == rank:
                        t = torch.from_numpy(o.get_occlusion(slot).copy())
                        ops.append(dist.P2POp(dist.isend, t, r))
                        keep.append(t)
                        sent[0] += 1
                    elif r == rank:
                        t = torch.empty(o.rows * o.cols, dtype=torch.float32)
                        ops.append(dist.P2POp(dist.irecv, t, owner))
                        landed.append((dl, t))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            for dl, t in landed:
                o.set_occlusion(dl, t.numpy())

        def all_gather(out, inp):
            parts = [torch.empty(n, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(parts, inp)
            out.copy_(torch.cat(parts))

        step = rdist.PeerShardedStep(None, n, cap, min_share=1, evaluate=evaluate, stage=stage, all_gather=all_gather, temperature=30.0)
        res = []
        for k, (_, frame) in enumerate(frames):
            o.set_observation(frame)
            ps = step.step(torch.from_numpy(poses[k][rank * n:(rank + 1) * n].copy()), uniforms[k])
            res.append((step.d_all.numpy().copy(), ps.numpy().copy()))
        q.put((rank, res, sent[0], step.counts.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_peer_step_matches_single_process(world):
    """dist.PeerShardedStep (global resampling, children in parent order, shared remote parents staged) on gloo
    ranks over oracle evaluators against ONE oracle holding all particles: the gathered log-likelihoods and the
    global parents of every step.  (On GPUs remote parents are read through mapped memory; here every one is
    staged, by message -- the index arithmetic under test is the same code.)"""
    ref = _peer_single(world)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 23500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_peer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sum(g[2] for g in got) > 0, "the scenario must stage planes across ranks"
    assert sum(g[3][0] for g in got) > 0
    for _, res, _, _ in got:
        for (ll_ref, ps_ref), (ll, ps) in zip(ref, res):
            assert np.abs(ll - ll_ref).max() <= 1e-9 * max(1.0, np.abs(ll_ref).max()), np.abs(ll - ll_ref).max()
            assert np.array_equal(ps, ps_ref)


def test_plan_shard_indices():
    """plan_shard by hand: 2 ranks x 4 particles, cap 8.  Sorted parents [1 1 2 5 | 5 5 6 7]: rank 0's child 3 has
    the remote parent 5 alone (read in place: global slot 1 * 8 + 1); rank 1's children are all local."""
    import torch
    ps = torch.tensor([1, 1, 2, 5, 5, 5, 6, 7])
    p0, s0, d0, c0 = rdist.plan_shard(ps, 4, 8, 0, 2)
    assert p0.tolist() == [1, 1, 2, 9] and d0.tolist() == [-1] * 4 and c0.tolist() == [1, 0, 0]
    p1, s1, d1, c1 = rdist.plan_shard(ps, 4, 8, 1, 2)
    assert p1.tolist() == [9, 9, 10, 11] and c1.tolist() == [0, 0, 0]
    # [0 0 0 0 | 0 0 3 6]: rank 1 has two children of rank 0's particle 0 -> staged once at local slot 4 (global 12),
    # one child of rank 0's particle 3 -> in place
    ps = torch.tensor([0, 0, 0, 0, 0, 0, 3, 6])
    p1, s1, d1, c1 = rdist.plan_shard(ps, 4, 8, 1, 2)
    assert p1.tolist() == [12, 12, 3, 10] and s1.tolist() == [0, -1, -1, -1] and d1.tolist() == [4, -1, -1, -1]
    assert c1.tolist() == [3, 2, 1]


class _FakeSensor:
    """What attach_peers needs of a sensor; `fail` = the step that raises on this rank."""

    def __init__(self, rank, fail=None):
        self.rank, self.fail, self.attached = rank, fail, None

    def ipc_export(self):
        if self.fail == "export":
            raise RuntimeError("export refused on rank %d" % self.rank)
        return bytes([self.rank]) * 512

    def ipc_attach(self, rank, blobs):
        if self.fail == "attach":
            raise RuntimeError("attach refused on rank %d" % self.rank)
        self.attached = [b[0] for b in blobs]


def _attach_worker(rank, world, port, fail_rank, fail, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = _FakeSensor(rank, fail if rank == fail_rank else None)
        try:
            rdist.attach_peers(s)
            q.put((rank, "ok", s.attached))
        except RuntimeError as e:
            q.put((rank, "raised", str(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fail", [None, "export", "attach"])
def test_attach_peers_fails_on_every_rank_alike(fail):
    """A rank that cannot export or attach must not leave the others waiting at a barrier: attach_peers raises the
    same RuntimeError on EVERY rank (bench.py --gpus N then falls back to shards with local parents)."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 25100 + (os.getpid() % 1500) + (0 if fail is None else len(fail))
    procs = [ctx.Process(target=_attach_worker, args=(r, world, port, 1, fail, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if fail is None:
        assert all(g[1] == "ok" and g[2] == [0, 1, 2] for g in got), got
    else:
        assert all(g[1] == "raised" and "rank 1" in g[2] and "refused" in g[2] for g in got), got


class _TrailSensor:
    """What PeerShardedStep's shared-trail policy needs of a sensor: the window fraction this rank sampled, and a record of the
    re-basings it was told to do."""

    def __init__(self, fractions):
        self.fractions, self.k, self.rebased = fractions, 0, []

    def window_fraction(self):
        return self.fractions[min(self.k, len(self.fractions) - 1)]

    def shared_trail_rebase(self, slot):
        self.rebased.append((self.k, slot))


def _trail_worker(rank, world, port, q):
    import torch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, cap, steps = 8, 16, 13
        # rank 1's windows grow past the threshold from step 6 on, rank 0's never do: BOTH must re-base, at the same steps
        s = _TrailSensor([0.02] * steps if rank == 0 else [0.02] * 6 + [0.4] * (steps - 6))

        def evaluate(poses, parent_idx, out):
            out.copy_(torch.arange(n, dtype=torch.float64) * 0.01 + rank)

        def all_gather(out, inp):
            parts = [torch.empty(n, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(parts, inp)
            out.copy_(torch.cat(parts))

        step = rdist.PeerShardedStep(s, n, cap, min_share=2, evaluate=evaluate, stage=lambda src, dst: None, all_gather=all_gather,
                                     temperature=1.0, shared_trail=True, trail_every=4, trail_threshold=0.10)
        g = torch.Generator().manual_seed(3)
        for k in range(steps):
            s.k = k
            step.step(torch.zeros(n, 12), torch.rand(n * world, dtype=torch.float64, generator=g))
        q.put((rank, s.rebased))
    finally:
        dist.destroy_process_group()


def test_peer_step_shared_trail_is_agreed_by_all_ranks():
    """dist.PeerShardedStep(shared_trail=True) on two gloo ranks: every trail_every steps the ranks all-reduce "my windows exceed the
    threshold"; when ANY rank's do, EVERY rank tells its handle to re-base on global slot 0 before the same step
    (rbs_shared_trail_rebase: the handles' shared planes stay identical only if all of them re-base in the same call)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 26300 + (os.getpid() % 1500)
    procs = [ctx.Process(target=_trail_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1] == [(8, 0), (12, 0)], got      # steps 4 (nobody above the threshold): nothing; 8 and 12: both ranks
