"""Particle sharding across the GPUs of one node: one process per GPU, contiguous
particle ranges, occlusion planes resident on the owning rank, one collective per sampling
block -- an all-gather of per-particle log-likelihoods (RCCL over xGMI with backend "nccl";
gloo in the CPU tests) -- and, after resampling, migration of only those parent planes whose
children could not be placed on the parent's own rank.

Everything here is host logic over torch.distributed; the evaluator is any object with
loglikes_poses / get_occlusion / set_occlusion (the product's RbSensor on GPUs; the CPU tests
plug the oracle in).  Each rank's sensor is created with 2 x shard slots: [0, shard) own
planes, [shard, 2*shard) staging for planes received from other ranks.
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
        ops, keep, landed = [], [], []
        npx = self.sensor.rows * self.sensor.cols
        on_device = self.device is not None and hasattr(self.sensor, "export_plane")
        stream = torch.cuda.current_stream().cuda_stream if on_device else None
        for src, dst, g in moves:
            if src == self.rank:
                if on_device:   # device-to-device out of the sensor, then RCCL straight from HBM
                    t = torch.empty(npx, dtype=torch.float32, device=self.device)
                    self.sensor.export_plane(g - self.lo, t.data_ptr(), stream)
                else:
                    t = torch.from_numpy(np.ascontiguousarray(self.sensor.get_occlusion(g - self.lo)))
                ops.append(dist.P2POp(dist.isend, t, dst, group=self.group))
                keep.append(t)
            elif dst == self.rank:
                t = torch.empty(npx, dtype=torch.float32, device=self.device if on_device else None)
                ops.append(dist.P2POp(dist.irecv, t, src, group=self.group))
                landed.append((self.staged[self.rank][g], t))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for slot, t in landed:
            if on_device:
                self.sensor.import_plane(slot, t.data_ptr(), stream)
            else:
                self.sensor.set_occlusion(slot, t.cpu().numpy())
        if on_device and landed:
            torch.cuda.current_stream().synchronize()   # staging tensors die with this scope
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
