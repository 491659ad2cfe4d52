"""Particle sharding across the GPUs of one node: one process per GPU, contiguous
particle ranges, occlusion planes resident on the owning rank, one collective per sampling
block -- an all-gather of per-particle log-likelihoods (RCCL over xGMI with backend "nccl";
gloo in the CPU tests) -- and, after resampling, migration of only those parent planes whose
children could not be placed on the parent's own rank.

Two forms.  ShardedSensor / ShardedRbSensor: host logic (numpy) over torch.distributed; the evaluator is
any object with loglikes_poses / get_occlusion / set_occlusion (the product's RbSensor on GPUs; the CPU
tests plug the oracle in).  Each rank's sensor is created with 2 x shard slots: [0, shard) own planes,
[shard, 2*shard) staging for planes received from other ranks; on GPUs a plane travels as its window
(rbs_export_window / rbs_import_window: rectangle + w x h values).
PeerShardedStep (round 4, what bench.py --gpus N times): the ranks' handles are attached to each other
(rbs_ipc_attach), parents on other ranks are read in place over xGMI, nothing migrates, the whole step is
device-resident torch arithmetic + library calls on one stream.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n, world):
    """Contiguous ranges: rank r owns [b[r], b[r+1])."""
    base, rem = divmod(n, world)
    sizes = [base + (1 if r < rem else 0) for r in range(world)]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def gather_loglikes(local_ll, bounds, group=None, device=None):
    """All-gather of the ranks' log-likelihood shards -> the full [N] vector on every rank."""
    world = dist.get_world_size(group)
    sizes = np.diff(bounds)
    m = int(sizes.max())
    t = torch.zeros(m, dtype=torch.float64, device=device)
    t[: len(local_ll)] = torch.as_tensor(local_ll, dtype=torch.float64, device=device)
    out = [torch.empty(m, dtype=torch.float64, device=device) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return np.concatenate([out[r][: sizes[r]].cpu().numpy() for r in range(world)])


def place_children(parents, bounds):
    """Deterministic child placement (identical on every rank).

    parents: global parent index of each of the N children (any order).
    Returns (child_rank[N], child_slot[N]): children are placed on their parent's rank while
    that rank has free slots (parent-affine), the surplus fills the remaining free slots of
    the other ranks in rank order."""
    parents = np.asarray(parents, dtype=np.int64)
    world = len(bounds) - 1
    n = len(parents)
    cap = np.diff(bounds).astype(np.int64)
    owner = np.searchsorted(bounds, parents, side="right") - 1
    child_rank = np.full(n, -1, dtype=np.int64)
    child_slot = np.full(n, -1, dtype=np.int64)
    # children in parent order (stable); the owners are then non-decreasing, so each rank's
    # children are one contiguous run: its first cap[r] stay, the rest are surplus
    order = np.argsort(parents, kind="stable")
    o_owner = owner[order]
    counts = np.bincount(o_owner, minlength=world).astype(np.int64)
    starts = np.cumsum(counts) - counts
    pos = np.arange(n, dtype=np.int64) - starts[o_owner]
    keep = pos < cap[o_owner]
    child_rank[order[keep]] = o_owner[keep]
    child_slot[order[keep]] = pos[keep]
    used = np.minimum(counts, cap)
    surplus = order[~keep]
    if len(surplus):
        free = cap - used
        ranks = np.repeat(np.arange(world, dtype=np.int64), free)
        first = np.cumsum(free) - free
        slots = np.arange(int(free.sum()), dtype=np.int64) - np.repeat(first, free) + np.repeat(used, free)
        child_rank[surplus] = ranks[: len(surplus)]
        child_slot[surplus] = slots[: len(surplus)]
    return child_rank, child_slot


class ShardedSensor:
    """Drives one rank's sensor inside a particle-sharded filter.

    Particles keep their GLOBAL ids (0..N-1: the row of the pose / weight arrays every rank
    holds identically).  Two maps are kept apart (they differ whenever a resampling is not
    followed at once by an updating call, e.g. between the sampling blocks of a multi-body
    tracker):

      layout[g]   the particle id EVALUATED at global slot g (rank-major: rank r owns global
                  slots bounds[r]..bounds[r+1]); recomputed identically on every rank after
                  each resampling, so no rank ever needs to ask where a particle is evaluated;
      inherit[j]  the PHYSICAL global slot holding the plane particle j inherits.  Planes only
                  move physically in an updating call (child j's posterior is written to the
                  slot j is evaluated at); between updating calls the own slots [0, shard) of
                  every rank are immutable, resamplings only compose `inherit`, and the planes a
                  rank needs from elsewhere are copies in its staging slots [stage0, 2*stage0),
                  remembered in `staged` (per destination rank, identical on every rank) until
                  the next updating call invalidates them."""

    def __init__(self, sensor, n_total, group=None, device=None):
        self.sensor, self.group, self.device = sensor, group, device
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_total = n_total
        self.bounds = shard_bounds(n_total, self.world)
        self.lo, self.hi = int(self.bounds[self.rank]), int(self.bounds[self.rank + 1])
        self.shard = self.hi - self.lo
        self.stage0 = int(np.diff(self.bounds).max())        # first staging slot; also the staging capacity
        self._fresh()

    def _fresh(self):
        self.layout = np.arange(self.n_total, dtype=np.int64)
        self.inherit = np.zeros(self.n_total, dtype=np.int64)   # after reset every particle inherits slot 0 ...
        self.slot_at_update = np.zeros(self.n_total, dtype=np.int64)
        self.staged = [dict() for _ in range(self.world)]
        self.local_parent_slots = np.zeros(self.shard, dtype=np.int32)
        self._after_reset = True     # ... of its OWN rank (reset fills every plane alike): no migration

    @property
    def owned(self):
        """Global particle ids this rank evaluates, in local slot order."""
        return self.layout[self.lo:self.hi]

    def reset(self):
        self.sensor.reset()
        self._fresh()

    def set_observation(self, image):
        self.sensor.set_observation(image)

    def loglikes(self, poses_all, update):
        """poses_all: [N, ...] poses of ALL particles by global id (identical on every rank).
        Evaluates this rank's particles and returns the full [N] log-likelihood vector in
        global-id order (one all-gather)."""
        idx = self.local_parent_slots.copy()
        ll = self.sensor.loglikes_poses(np.asarray(poses_all)[self.owned], idx, update=update)
        if update:
            # slot k of this rank now physically holds the plane of particle owned[k]
            self.local_parent_slots = np.arange(self.shard, dtype=np.int32)
            slot_of = np.empty(self.n_total, dtype=np.int64)
            slot_of[self.layout] = np.arange(self.n_total)
            self.inherit = slot_of
            self.slot_at_update = slot_of.copy()
            self.staged = [dict() for _ in range(self.world)]
            self._after_reset = False
        by_slot = gather_loglikes(ll, self.bounds, self.group, self.device)
        out = np.empty(self.n_total)
        out[self.layout] = by_slot
        return out

    def resample(self, parents):
        """parents[j] = global id of the particle child j inherits from (identical on all
        ranks).  Returns the planes moved between ranks as (src, dst, global slot) triples."""
        return self._apply(self.inherit[np.asarray(parents, dtype=np.int64)])

    def set_inheritance(self, plane_ids):
        """plane_ids[j] = id, AT THE LAST UPDATING CALL, of the particle whose plane particle j
        inherits (the meaning of RbSensor::loglikes' `indices`)."""
        return self._apply(self.slot_at_update[np.asarray(plane_ids, dtype=np.int64)])

    def _apply(self, inherit):
        """Places the particles parent-affine for the physical slots `inherit`, copies the planes
        of parents whose children landed on another rank into that rank's staging slots (unless
        a copy made since the last updating call is still there), and updates the layout:
        afterwards particle j is evaluated where (a copy of) its plane is."""
        self.inherit = inherit
        if self._after_reset:
            # every plane of every rank is the initial plane: particles stay where they are and
            # inherit the first slot of their own rank
            self.local_parent_slots = np.zeros(self.shard, dtype=np.int32)
            return []
        pslot = inherit
        child_rank, child_slot = place_children(pslot, self.bounds)
        owner = np.searchsorted(self.bounds, pslot, side="right") - 1
        away = owner != child_rank
        moves = []
        if away.any():
            need = np.unique(np.stack([owner[away], child_rank[away], pslot[away]], axis=1), axis=0)
            for dst in np.unique(need[:, 1]):
                rows = need[need[:, 1] == dst]
                cache = self.staged[int(dst)]
                new = [r for r in rows if int(r[2]) not in cache]
                if len(cache) + len(new) > self.stage0:     # staging full: start over with what is needed now
                    cache.clear()
                    new = list(rows)
                for r in new:
                    cache[int(r[2])] = self.stage0 + len(cache)
                    moves.append((int(r[0]), int(dst), int(r[2])))
        npx = self.sensor.rows * self.sensor.cols
        on_device = self.device is not None and hasattr(self.sensor, "export_window")
        if on_device:
            self._migrate_windows(moves)
        else:
            ops, keep, landed = [], [], []
            for src, dst, g in moves:
                if src == self.rank:
                    t = torch.from_numpy(np.ascontiguousarray(self.sensor.get_occlusion(g - self.lo)))
                    ops.append(dist.P2POp(dist.isend, t, dst, group=self.group))
                    keep.append(t)
                elif dst == self.rank:
                    t = torch.empty(npx, dtype=torch.float32)
                    ops.append(dist.P2POp(dist.irecv, t, src, group=self.group))
                    landed.append((self.staged[self.rank][g], t))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            for slot, t in landed:
                self.sensor.set_occlusion(slot, t.numpy())
        new_layout = np.empty(self.n_total, dtype=np.int64)
        new_layout[self.bounds[child_rank] + child_slot] = np.arange(self.n_total)
        self.layout = new_layout
        mine = self.owned
        g = pslot[mine]
        slots = (g - self.lo).astype(np.int32)
        remote = owner[mine] != self.rank
        if remote.any():
            cache = self.staged[self.rank]
            slots[remote] = np.array([cache[int(x)] for x in g[remote]], dtype=np.int32)
        self.local_parent_slots = slots
        return moves


    def _migrate_windows(self, moves):
        """Device path (one process per GPU, RCCL): a plane travels as its WINDOW -- rectangle + w x h values,
        about 3 % of rows x cols -- straight out of / into HBM (rbs_export_window / rbs_import_window).  Two
        rounds of point-to-point transfers: the rectangles (16 bytes each; the receiver needs them to size its
        buffers), then the payloads.  (Whole planes, round 3: 1.2 MB per move at 640x480.)"""
        stream = torch.cuda.current_stream().cuda_stream
        out, inc = [], []          # (dst, rect tensor, payload tensor), (src, slot, rect tensor)
        for src, dst, g in moves:
            if src == self.rank:
                x0, y0, x1, y1 = self.sensor.get_window(g - self.lo)
                area = max(0, x1 - x0) * max(0, y1 - y0)
                pay = torch.empty(max(area, 1), dtype=torch.float32, device=self.device)
                rect = self.sensor.export_window(g - self.lo, pay.data_ptr(), pay.numel(), stream)
                out.append((dst, torch.tensor(rect, dtype=torch.int32, device=self.device), pay[:max(area, 1)]))
            elif dst == self.rank:
                inc.append((src, self.staged[self.rank][g], torch.empty(4, dtype=torch.int32, device=self.device)))
        ops = [dist.P2POp(dist.isend, r, d, group=self.group) for d, r, _ in out]
        ops += [dist.P2POp(dist.irecv, r, s_, group=self.group) for s_, _, r in inc]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        rects = [tuple(int(v) for v in r.cpu()) for _, _, r in inc]
        pays = [torch.empty(max(1, max(0, r[2] - r[0]) * max(0, r[3] - r[1])), dtype=torch.float32, device=self.device) for r in rects]
        ops = [dist.P2POp(dist.isend, p_, d, group=self.group) for d, _, p_ in out]
        ops += [dist.P2POp(dist.irecv, p_, s_, group=self.group) for (s_, _, _), p_ in zip(inc, pays)]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for (_, slot, _), r, p_ in zip(inc, rects, pays):
            self.sensor.import_window(slot, r, p_.data_ptr(), stream)
        self.window_floats_moved = getattr(self, "window_floats_moved", 0) + sum(int(p_.numel()) for _, _, p_ in out)
        if inc or out:
            torch.cuda.current_stream().synchronize()   # the transfer tensors die with this scope


class ShardedRbSensor:
    """A ShardedSensor behind the single-sensor interface the trackers drive
    (set_observation / loglikes_poses(poses, indices, update) / reset), so that
    `ParticleTracker(transition, ShardedRbSensor(...), ...)` run identically on every rank
    (same seed) is the particle-sharded tracker: every rank holds all particle STATES (they are
    96 bytes each), evaluates its shard, all-gathers the log-likelihoods and migrates planes.

    `indices[i]` is, as for a single sensor, the plane particle i inherits, named by the id the
    plane's particle had at the last updating call."""

    def __init__(self, sensor, n_total, group=None, device=None):
        self.ss = ShardedSensor(sensor, n_total, group, device)
        self.n = n_total
        self.rows, self.cols = sensor.rows, sensor.cols
        self.applied = np.arange(n_total, dtype=np.int64)   # plane (by last-update id) each particle holds now
        self.moves = 0

    def reset(self):
        self.ss.reset()
        self.applied = np.arange(self.n, dtype=np.int64)

    def set_observation(self, image):
        self.ss.set_observation(image)

    def loglikes_poses(self, poses, indices, update=False):
        want = np.asarray(indices, dtype=np.int64)
        if len(want) != self.n:
            raise ValueError("a sharded sensor evaluates all particles in every call")
        if not np.array_equal(want, self.applied):
            self.moves += len(self.ss.set_inheritance(want))
            self.applied = want.copy()
        ll = self.ss.loglikes(poses, update)
        if update:
            self.applied = np.arange(self.n, dtype=np.int64)
            indices[:] = np.arange(self.n, dtype=indices.dtype)
        return ll

    def close(self):
        self.ss.sensor.close()


# ------------------------------------------------------------------------------------------------
# One process per GPU with the other ranks' planes MAPPED (rbs_ipc_attach): no plane migrates.
# Device-agnostic torch code (the CPU tests run the same functions on CPU tensors).

def global_resample(ll_all, uniforms, temperature=1.0):
    """Multinomial resampling over ALL ranks' particles (upper_bound of the cumulative normalised
    weights at host-supplied uniforms), identical on every rank: the parent of each of the N children as an index
    into the gathered log-likelihood vector, SORTED -- children are exchangeable, and in parent order the children
    of rank r's particles are one contiguous run that mostly coincides with rank r's slots [r n, (r + 1) n):
    slot g simply evaluates child g (a stable n-way partition without any bookkeeping)."""
    w = torch.exp((ll_all - ll_all.max()) / temperature)   # (temperature > 1: flatter weights -- synthetic workloads only)
    c = torch.cumsum(w, 0)
    c = c / c[-1]
    p = torch.searchsorted(c, uniforms, right=True).clamp_(max=ll_all.numel() - 1)
    return torch.sort(p).values


def plan_shard(parents_sorted, n, cap, rank, min_share=2):
    """What rank `rank` does with its slice of the sorted parents (fixed-size tensor arithmetic, no host
    synchronisation).  A parent on another rank is read IN PLACE by its child (over xGMI) -- unless at least
    `min_share` of this rank's children share it: then its window is pulled once into a local staging slot
    n + j and the children read that.  Returns
      parent_idx [n] int32   global slot (rank_of_owner * cap + local slot) each local child inherits from,
      stage_src / stage_dst [n] int32   entry i staged iff stage_dst[i] >= 0 (rbs_stage_windows),
      counts [3] int64       children with a remote parent, of them served from staging, planes staged."""
    dev = parents_sorted.device
    mine = parents_sorted[rank * n:(rank + 1) * n]
    owner = torch.div(mine, n, rounding_mode="floor")
    remote = owner != rank
    pg = owner * cap + (mine - owner * n)
    new = torch.ones(n, dtype=torch.bool, device=dev)
    new[1:] = mine[1:] != mine[:-1]
    run = torch.cumsum(new.to(torch.int64), 0) - 1
    runlen = torch.zeros(n, dtype=torch.int64, device=dev).scatter_add_(0, run, torch.ones(n, dtype=torch.int64, device=dev))
    shared = remote & (runlen[run] >= min_share)
    start = new & shared
    sidx = torch.cumsum(start.to(torch.int64), 0) - 1
    parent_idx = torch.where(shared, rank * cap + n + sidx, pg).to(torch.int32)
    neg = torch.full((n,), -1, dtype=torch.int64, device=dev)
    stage_src = torch.where(start, pg, neg).to(torch.int32)
    stage_dst = torch.where(start, n + sidx, neg).to(torch.int32)
    counts = torch.stack([remote.sum(), shared.sum(), start.sum()])
    return parent_idx, stage_src, stage_dst, counts


def attach_peers(sensor, group=None):
    """Exchange the ranks' rbs_ipc_export blobs (an all-gather of 512 bytes each) and attach.  A failure on any
    rank (export or attach) is raised on EVERY rank, after all of them have left the hand-shake: nobody is left
    waiting at a barrier for a rank that has given up."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    err = None
    try:
        mine = sensor.ipc_export()
    except Exception as e:   # noqa: BLE001 -- reported to every rank below
        mine, err = None, e
    blobs = [None] * world
    dist.all_gather_object(blobs, mine, group=group)
    usable = all(b is not None for b in blobs)
    # one rank at a time: two processes importing each other's large buffers at the same moment were seen to
    # block each other inside hipIpcOpenMemHandle for ever (3 GB slabs, ROCm 7.2)
    for r in range(world):
        if r == rank and usable and err is None:
            try:
                sensor.ipc_attach(rank, blobs)
            except Exception as e:   # noqa: BLE001
                err = e
        dist.barrier(group=group)
    said = [None] * world
    dist.all_gather_object(said, None if err is None else repr(err), group=group)
    bad = [f"rank {r}: {m}" for r, m in enumerate(said) if m]
    if bad:
        raise RuntimeError("attach_peers failed -- " + "; ".join(bad))


class PeerShardedStep:
    """The filter step of SURVEY 8(e) across processes, everything on one stream, no host synchronisation:
        loglikes(update) on this rank's n particles (parents = global slots; remote ones read in place)
        -> rbs_stream_join -> all-gather of the log-likelihoods (RCCL) -> global_resample -> plan_shard
        -> rbs_stage_windows (shared remote parents pulled once) -> the next step's parent indices.
    The sensor has max_particles = cap >= 2 n (n own slots + staging) and is attached (attach_peers).
    `evaluate(poses, parent_idx, out)` / `stage(src, dst)` / `all_gather(out, inp)` replace the three device
    operations (the CPU tests drive the oracle through them; the one-GPU test gathers through gloo).
    fused=True: global_resample + plan_shard as ONE library kernel (rbs_peer_resample) instead of ~45 tensor
    kernels; the step's uniforms must then be SORTED ascending (children are exchangeable: the parents come out
    sorted either way, and they are the same parents), step() returns this rank's slice of the sorted parents.
    shared_trail=True (round 6): the ranks' planes are stored against ONE shared background plane once windows have grown
    (include/rbsensor_mi355x.h, rbs_shared_trail_rebase): every trail_every steps each rank looks at the window fraction its
    handle sampled last (no synchronisation), the ranks agree by ONE all-reduce (max) of that flag -- `agree(flag) -> bool`
    replaces it -- and, if any rank's windows exceed trail_threshold of the frame, every rank tells its handle to re-base on
    global slot 0 at the step that follows.  Values do not change by a bit; stored windows shrink to what the particles do not
    share with their common ancestor."""

    def __init__(self, sensor, n, cap, group=None, device=None, min_share=2, stream=None, evaluate=None, stage=None, all_gather=None,
                 temperature=1.0, fused=False, world=None, rank=None, shared_trail=False, trail_every=32, trail_threshold=0.10, agree=None):
        self.sensor, self.n, self.cap, self.group, self.device, self.min_share = sensor, n, cap, group, device, min_share
        self.temperature = temperature
        self.fused = fused
        # (world / rank given: no process group is consulted -- world = 1 is the same step on ONE rank, its "all-gather" a copy:
        # what bench.py's single-GPU line times beside the headline so that the multi-rank values have a like-for-like reference)
        self.world, self.rank = (world, rank or 0) if world is not None else (dist.get_world_size(group), dist.get_rank(group))
        self.N = self.world * n
        # ONE stream carries the whole step: the library calls (loglikes_device, stream_join, peer_resample,
        # stage_windows) and torch's operations (the all-gather, global_resample / plan_shard) must be ordered among
        # themselves -- the all-gather reads what the raster kernel writes, the next step reads the plan.  On a device
        # the default is therefore torch's CURRENT stream (what torch's own operations are issued on), never the
        # handle's private stream; a caller that passes a stream must run its torch operations under it.
        if stream is None and device is not None and torch.device(device).type == "cuda":
            stream = torch.cuda.current_stream(torch.device(device)).cuda_stream
        self.stream = stream
        self.d_out = torch.zeros(n, dtype=torch.float64, device=device)
        self.d_all = torch.zeros(self.N, dtype=torch.float64, device=device)
        self.parent_idx = torch.full((n,), self.rank * cap, dtype=torch.int32, device=device)   # after reset: any own slot
        self.counts = torch.zeros(4 if fused else 3, dtype=torch.int64, device=device)
        self.children = 0
        self._evaluate = evaluate or self._evaluate_device
        self._stage = stage or (lambda src, dst: sensor.stage_windows(src.data_ptr(), dst.data_ptr(), n, self.stream))
        self._all_gather = all_gather or ((lambda out, inp: out.copy_(inp)) if self.world == 1 and world is not None
                                          else (lambda out, inp: dist.all_gather_into_tensor(out, inp, group=group)))
        self._keep = None
        self.shared_trail, self.trail_every, self.trail_threshold, self._steps = shared_trail, max(1, int(trail_every)), trail_threshold, 0
        self._agree = agree or self._agree_all_reduce
        if fused:   # two sets of plan buffers: a step's plan is read by kernels enqueued behind the next step's
            self._plans = [tuple(torch.full((n,), -1, dtype=torch.int32, device=device) for _ in range(4)) for _ in range(2)]
            self._flip = 0

    def _evaluate_device(self, poses, parent_idx, out):
        self.sensor.loglikes_device(poses.data_ptr(), parent_idx.data_ptr(), self.n, True, out.data_ptr(), self.stream)
        self.sensor.stream_join(self.stream)

    def _agree_all_reduce(self, flag):
        """True on every rank if `flag` is true on any: one all-reduce (max) of one number (the backend's device)."""
        if self.world == 1:
            return bool(flag)
        on_device = dist.get_backend(self.group) == "nccl"
        t = torch.tensor([1.0 if flag else 0.0], device=self.device if on_device else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(t.item() > 0.5)

    def step(self, poses, uniforms):
        """One updating call + exchange + resampling.  poses: this rank's [n, 12 bodies] tensor; `uniforms` [N]
        identical on every rank (fused: ascending).  Returns the sorted global parents (indices into the gathered
        vector) -- fused: this rank's n of them."""
        if self.shared_trail and self._steps > 0 and self._steps % self.trail_every == 0:
            if self._agree(self.sensor.window_fraction() > self.trail_threshold):
                self.sensor.shared_trail_rebase(0)      # (every rank, the same slot, before the same step)
        self._steps += 1
        self._evaluate(poses, self.parent_idx, self.d_out)
        self._all_gather(self.d_all, self.d_out)
        if self.fused:
            parent_idx, src, dst, mine = self._plans[self._flip]
            self._flip ^= 1
            self.sensor.peer_resample(self.d_all.data_ptr(), uniforms.data_ptr(), self.N, self.n, self.rank, self.min_share,
                                      self.temperature, parent_idx.data_ptr(), src.data_ptr(), dst.data_ptr(), mine.data_ptr(),
                                      self.counts.data_ptr(), self.stream)
            self._stage(src, dst)
            self._keep = (self.parent_idx, poses, uniforms)
            self.parent_idx = parent_idx
            self.last_parents = mine
            self.children += self.n
            return mine
        ps = global_resample(self.d_all, uniforms, self.temperature)
        self.last_parents = ps
        parent_idx, src, dst, counts = plan_shard(ps, self.n, self.cap, self.rank, self.min_share)
        self._stage(src, dst)
        self._keep = (self.parent_idx, src, dst, poses)   # (alive until the kernels reading them are enqueued behind the next step's)
        self.parent_idx = parent_idx
        self.counts += counts
        self.children += self.n
        return ps
