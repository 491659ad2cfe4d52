"""Round 5: the plugin-surface call (deltas composed on the device), the split launch (geometry kernel ->
depth tiles -> likelihood kernel), and the entry-point corners ADVICE r4 named."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
import scenarios as sc
from dbot_ros_amd import RbSensor, RbSensorError, _capi, synth  # noqa: F401

pytestmark = pytest.mark.gpu

TOL_EAGER = 1e-9


def rel_err(a, b):
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


def _deltas(rng, n, parts, scale=1.0):
    d = np.zeros((n, parts, 12))
    d[:, :, 0:3] = rng.normal(0.0, 0.004 * scale, (n, parts, 3))
    d[:, :, 3:6] = rng.normal(0.0, 0.03 * scale, (n, parts, 3))
    d[:, :, 6:12] = rng.normal(0.0, 1.0, (n, parts, 6))    # velocities: must be ignored
    d[0, :, 3:6] = 0.0                                       # the zero rotation (the series branch of sin(x)/x)
    return d.reshape(n, parts * 12)


@pytest.mark.parametrize("meshes,cols,rows", [(("m1",), 640, 480), (("m1_l2", "box12"), 320, 240)])
def test_loglikes_deltas_composes_on_the_device(gpu_lib, meshes, cols, rows):
    """rbs_loglikes_deltas = the filter's own arguments (state deltas + default poses,
    R:source/dbot_ros/object_tracker_ros.hpp:49): the device's compositions agree with
    oracle/tracker_oracle.c orc_compose_poses to the last bits (sin / cos / sqrt are the device library's),
    and the log-likelihoods are those of the oracle fed with exactly the poses the device evaluated."""
    parts, n = len(meshes), 48
    om, cam, P = sc.make_scene(meshes, cols, rows, max_particles=n)
    o = ob.Oracle(om, cam, P, max_particles=n, mode=ob.EAGER)
    frames = sc.make_frames(o, parts, 3, seed=3)
    rng = np.random.default_rng(11)
    with RbSensor(om, cam, P, max_particles=n, precision="f64") as g:
        g.reset(); o.reset()
        ig, io = np.zeros(n, np.int32), np.zeros(n, np.int32)
        for k, (truth, frame) in enumerate(frames):
            # default pose = the truth as position + rotation vector (one body turned by almost pi: the large-angle branch)
            from dbot_ros_amd import pose as ps
            dflt = np.zeros((parts, 12))
            for b in range(parts):
                dflt[b, 0:3] = truth[b, 9:12]
                dflt[b, 3:6] = ps.matrix_to_rotvec(truth[b, :9].reshape(3, 3))
            g.integrated_poses = dflt.reshape(-1).copy()
            d = _deltas(rng, n, parts, scale=1.0 + k)
            g.set_observation(frame); o.set_observation(frame)
            ll = g.loglikes_deltas(d, ig, update=True)
            got = g.get_poses(n)
            want = ob.compose_poses(d, dflt, parts)
            assert np.abs(got - want).max() <= 4e-16 * max(1.0, np.abs(want).max()) * 4, np.abs(got - want).max()
            ref = o.loglikes_poses(got, io, update=True)
            assert rel_err(ll, ref).max() <= TOL_EAGER, rel_err(ll, ref).max()
            # ... and next to the host composition the Python mirror performs: the same numbers to 1e-9
            assert (ig == np.arange(n)).all()
            w = np.exp(ll - ll.max())
            ig = np.sort(rng.choice(n, size=n, p=w / w.sum())).astype(np.int32)
            io = ig.copy()


def test_loglikes_deltas_packed_stride_and_bad_arguments(gpu_lib):
    n = 8
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=n)
    lib = _capi.load()
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    rng = np.random.default_rng(2)
    with RbSensor(om, cam, P, max_particles=n, precision="f64") as g:
        truth = synth.truth_pose(1)
        frame = synth.make_frame(g.render_depth(truth), 120, 160, rng)
        from dbot_ros_amd import pose as ps
        dflt12 = np.zeros(12); dflt12[0:3] = truth[0, 9:12]; dflt12[3:6] = ps.matrix_to_rotvec(truth[0, :9].reshape(3, 3))
        g.integrated_poses = dflt12.copy()
        d12 = _deltas(rng, n, 1)
        g.reset(); g.set_observation(frame)
        idx = np.zeros(n, np.int32)
        a = g.loglikes_deltas(d12, idx.copy(), update=False)
        # the same deltas packed six per body
        d6 = np.ascontiguousarray(d12.reshape(n, 12)[:, :6]); f6 = np.ascontiguousarray(dflt12[:6])
        out = np.empty(n)
        rc = lib.rbs_loglikes_deltas(g._h, d6.ctypes.data_as(dp), f6.ctypes.data_as(dp), 6, idx.ctypes.data_as(ip), n, 0, out.ctypes.data_as(dp))
        assert rc == 0 and np.array_equal(a, out)
        rc = lib.rbs_loglikes_deltas(g._h, d6.ctypes.data_as(dp), f6.ctypes.data_as(dp), 5, idx.ctypes.data_as(ip), n, 0, out.ctypes.data_as(dp))
        assert rc == _capi.RBS_ERR_INVALID_ARGUMENT
        rc = lib.rbs_loglikes_deltas(g._h, None, f6.ctypes.data_as(dp), 6, idx.ctypes.data_as(ip), n, 0, out.ctypes.data_as(dp))
        assert rc == _capi.RBS_ERR_INVALID_ARGUMENT
        # a handle over several shards takes the same call
    with RbSensor(om, cam, P, max_particles=n, precision="f64", device_ids=[0, 0]) as grp:
        grp.integrated_poses = dflt12.copy()
        grp.reset(); grp.set_observation(frame)
        b = grp.loglikes_deltas(d12, idx.copy(), update=False)
        assert np.array_equal(a, b)


def test_tail_weight_zero_is_accepted_and_evaluated_at_the_floor(gpu_lib):
    """the reference hands kinect/tail_weight through unchecked, 0 included.  The library accepts 0 and
    evaluates it as RBS_TAIL_WEIGHT_FLOOR = 1e-9 (include/rbsensor_mi355x.h says why)."""
    n = 32
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=n)
    from dbot_ros_amd import RbSensorBuilder
    P0 = RbSensorBuilder.Parameters(sample_count=n); P0.kinect.tail_weight = 0.0
    Pf = RbSensorBuilder.Parameters(sample_count=n); Pf.kinect.tail_weight = 1e-9
    o = ob.Oracle(om, cam, Pf, max_particles=n, mode=ob.EAGER)
    frames = sc.make_frames(o, 1, 3, seed=4)
    ref = sc.run_sequence(o, frames, n)
    with RbSensor(om, cam, P0, max_particles=n, precision="f64") as g:
        got = sc.run_sequence(g, frames, n)
    for a, b in zip(got, ref):
        assert np.isfinite(a).all()
        assert rel_err(a, b).max() <= TOL_EAGER, rel_err(a, b).max()


def test_prefetch_is_refused_before_anything_is_enqueued(gpu_lib):
    """a second rbs_loglikes_prefetch before rbs_set_observation_prefetched used to fail AFTER the updating call's
    kernels had been enqueued (planes flipped, indices stale).  It is refused up front: nothing changes."""
    n = 16
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=n)
    o = ob.Oracle(om, cam, P, max_particles=n, mode=ob.EAGER)
    frames = sc.make_frames(o, 1, 3, seed=9)
    rng = np.random.default_rng(1)
    poses = [synth.particle_poses(t, n, rng, scale=1.5) for t, _ in frames]
    with RbSensor(om, cam, P, max_particles=n, precision="f64") as g, RbSensor(om, cam, P, max_particles=n, precision="f64") as plain:
        for s in (g, plain):
            s.reset(); s.set_observation(frames[0][1])
        idx = np.zeros(n, np.int32)
        a0 = g.loglikes_poses_prefetch(poses[0], idx, frames[1][1], update=True)
        b0 = plain.loglikes_poses(poses[0], np.zeros(n, np.int32), update=True)
        assert np.array_equal(a0, b0)
        parents = np.sort(rng.integers(0, n, n)).astype(np.int32)
        idx = parents.copy()
        with pytest.raises(RbSensorError) as e:     # the frame uploaded ahead has not been installed
            g.loglikes_poses_prefetch(poses[1], idx, frames[2][1], update=True)
        assert e.value.code == _capi.RBS_ERR_INVALID_ARGUMENT
        assert np.array_equal(idx, parents)          # nothing ran: the indices are the caller's
        g.set_observation_prefetched()
        a1 = g.loglikes_poses(poses[1], idx, update=True)
        plain.set_observation(frames[1][1])
        b1 = plain.loglikes_poses(poses[1], parents.copy(), update=True)
        assert np.array_equal(a1, b1)               # ... and the planes are those of the first call


def test_a_prefetched_frame_is_abandoned_by_set_observation_device(gpu_lib):
    """rbs_set_observation_device (and rbs_reset) between a prefetch and its installation abandon the frame;
    rbs_set_observation_prefetched then fails instead of re-installing a stale image."""
    import torch
    n = 8
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=n)
    rng = np.random.default_rng(1)
    with RbSensor(om, cam, P, max_particles=n, precision="f64") as g:
        truth = synth.truth_pose(1)
        f0 = synth.make_frame(g.render_depth(truth), 120, 160, rng)
        f1 = synth.make_frame(g.render_depth(truth), 120, 160, rng)
        poses = synth.particle_poses(truth, n, rng)
        for breaker in ("device", "reset"):
            g.reset(); g.set_observation(f0)
            g.loglikes_poses_prefetch(poses, np.zeros(n, np.int32), f1, update=True)
            if breaker == "device":
                d = torch.from_numpy(f0).to("cuda:0")
                torch.cuda.synchronize()
                g.set_observation_device(d.data_ptr())
            else:
                g.reset()
            with pytest.raises(RbSensorError) as e:
                g.set_observation_prefetched()
            assert e.value.code == _capi.RBS_ERR_INVALID_ARGUMENT
            g.synchronize()


def test_import_window_enlarges_the_slabs(gpu_lib):
    """a window exported by a handle whose slabs have grown is imported by one whose slabs are still
    at their initial size -- the receiver's slabs grow (as behind rbs_import_plane), the plane arrives whole."""
    import torch
    n = 8
    om, cam, P = sc.make_scene(("m1",), 640, 480, max_particles=n)
    rng = np.random.default_rng(3)
    with RbSensor(om, cam, P, max_particles=n, precision="f64", slab_px=16384) as src, \
            RbSensor(om, cam, P, max_particles=n, precision="f64", slab_px=1024) as dst:
        truth = synth.truth_pose(1, z=0.45)                     # a near object: regions of ~15 000 px
        frame = synth.make_frame(src.render_depth(truth), 480, 640, rng)
        src.reset(); src.set_observation(frame)
        src.loglikes_poses(synth.particle_poses(truth, n, rng), np.zeros(n, np.int32), update=True)
        win = src.get_window(2)
        w, h = win[2] - win[0], win[3] - win[1]
        assert w * h > 1024
        buf = torch.empty(w * h, dtype=torch.float32, device="cuda:0")
        rect = src.export_window(2, buf.data_ptr(), buf.numel())
        torch.cuda.synchronize()
        dst.reset(); dst.set_observation(frame)
        dst.loglikes_poses(synth.particle_poses(truth, n, rng) + 0.0, np.zeros(n, np.int32), update=False)
        dst.import_window(5, rect, buf.data_ptr())
        dst.synchronize()
        assert tuple(dst.get_window(5)) == tuple(win)
        # same background level on both sides? no: src stepped once.  Compare inside the window only.
        a, b = src.get_occlusion(2).reshape(480, 640), dst.get_occlusion(5).reshape(480, 640)
        assert np.array_equal(a[win[1]:win[3], win[0]:win[2]], b[win[1]:win[3], win[0]:win[2]])


# ---------------------------------------------------------------- split launch
@pytest.mark.parametrize("meshes,cols,rows,n,slab", [(("m1",), 640, 480, 64, 0), (("m1",), 640, 480, 64, 38400),
                                                     (("m1", "m2", "m3"), 640, 480, 24, 0), (("m4",), 1280, 960, 12, 0),
                                                     (("m1_l2",), 322, 241, 40, 0)])
def test_split_launch_matches_the_oracle(gpu_lib, monkeypatch, meshes, cols, rows, n, slab):
    """RBS_SPLIT=1: geometry kernel -> depth tiles in memory -> likelihood kernel.  The same bars as the one-kernel launch:
    log-likelihoods vs the device-rule oracle 1e-9, planes bit-exact but for 1-ulp posteriors, windows identical to the
    one-kernel launch's, read-only calls included."""
    nb = len(meshes)
    om, cam, P = sc.make_scene(meshes, cols, rows, max_particles=n)
    o = ob.Oracle(om, cam, P, max_particles=n, mode=ob.EAGER)
    frames = sc.make_frames(o, nb, 4, seed=5, z=0.5 if "m4" in meshes else 0.7)
    ref = sc.run_sequence(o, frames, n, n_bodies=nb)
    monkeypatch.setenv("RBS_SPLIT", "0")
    with RbSensor(om, cam, P, max_particles=n, precision="f64", slab_px=slab) as mono:
        base = sc.run_sequence(mono, frames, n, n_bodies=nb)
        monkeypatch.setenv("RBS_SPLIT", "1")
        monkeypatch.setenv("RBS_SPLIT_ITEMS_PER_PARTICLE", "400")   # (few particles: a rectangle is cut into many row bands)
        with RbSensor(om, cam, P, max_particles=n, precision="f64", slab_px=slab) as g:
            got = sc.run_sequence(g, frames, n, n_bodies=nb)
            for a, b, c in zip(got, ref, base):
                assert rel_err(a, b).max() <= TOL_EAGER, rel_err(a, b).max()
                assert rel_err(a, c).max() <= 1e-12          # (the tile size differs: items, hence the order of partial sums, may)
            for slot in (0, n // 2, n - 1):
                assert g.get_window(slot) == mono.get_window(slot)
                pg, po = g.get_occlusion(slot), o.get_occlusion(slot)
                diff = pg != po
                assert diff.mean() <= 1e-4
            rng = np.random.default_rng(8)
            poses = synth.particle_poses(frames[-1][0], n, rng, scale=2.0)
            idx = rng.integers(0, n, n).astype(np.int32)
            ro = g.loglikes_poses(poses, idx.copy(), update=False)
            assert rel_err(ro, mono.loglikes_poses(poses, idx.copy(), update=False)).max() <= 1e-12
            assert rel_err(ro, o.loglikes_poses(poses, idx.copy(), update=False)).max() <= TOL_EAGER


def test_split_launch_contains_items_beyond_its_buffer(gpu_lib, monkeypatch):
    """The hand-over buffer holds RBS_SPLIT_ITEMS_PER_PARTICLE x n (+ 1 024) tiles; an item beyond it is contained
    (its particle's log-likelihood is NaN), never a wild access."""
    n = 2000
    om, cam, P = sc.make_scene(("m1_l2",), 640, 480, max_particles=n)
    monkeypatch.setenv("RBS_SPLIT", "1")
    monkeypatch.setenv("RBS_SPLIT_ITEMS_PER_PARTICLE", "1")
    rng = np.random.default_rng(0)
    with RbSensor(om, cam, P, max_particles=n, precision="f64") as g:
        truth = synth.truth_pose(1, z=0.12)          # fills the image: ~40 tiles per particle
        frame = synth.make_frame(g.render_depth(truth), 480, 640, rng)
        g.reset(); g.set_observation(frame)
        ll = g.loglikes_poses(np.repeat(truth[None], n, 0), np.zeros(n, np.int32), update=True)
        assert np.isnan(ll).any() and np.isfinite(ll).any()
        g.synchronize()


def test_borrowed_frame_gives_the_same_bits(gpu_lib):
    """rbs_set_observation_borrowed: the frame is staged by the next likelihood call between its geometry and its likelihood
    kernel (two-kernel launch) -- log-likelihoods and planes are bit for bit those of rbs_set_observation + the one-kernel
    launch, on a resampled sequence; a borrowed frame that another frame overtakes is dropped, one that rbs_synchronize or
    a device-pointer call meets is staged there."""
    import torch
    n = 96
    om, cam, P = sc.make_scene(("m1",), 640, 480, max_particles=n)
    o = ob.Oracle(om, cam, P, max_particles=n, mode=ob.EAGER)
    frames = sc.make_frames(o, 1, 4, seed=21)
    rng = np.random.default_rng(4)
    poses = [synth.particle_poses(t, n, rng, scale=1.5) for t, _ in frames]
    parents = [np.sort(rng.integers(0, n, n)).astype(np.int32) for _ in frames]
    with RbSensor(om, cam, P, max_particles=n, precision="f64") as a, RbSensor(om, cam, P, max_particles=n, precision="f64") as b:
        a.reset(); b.reset()
        ia, ib = np.zeros(n, np.int32), np.zeros(n, np.int32)
        for k, (_, frame) in enumerate(frames):
            f64 = frame.astype(np.float64)
            a.set_observation(f64)
            if k == 2:      # overtaken: dropped, but its clock tick stays -- the same on both sides
                a.set_observation(f64)
                b.set_observation_borrowed(np.full_like(f64, 0.3))
            b.set_observation_borrowed(f64)
            la = a.loglikes_poses(poses[k], ia, update=True)
            lb = b.loglikes_poses(poses[k], ib, update=True)
            assert np.array_equal(la, lb), np.abs(la - lb).max()
            ia, ib = parents[k].copy(), parents[k].copy()
        for slot in (0, n // 3, n - 1):
            assert a.get_window(slot) == b.get_window(slot)
            assert np.array_equal(a.get_occlusion(slot), b.get_occlusion(slot))
        # the driver's float pixels, borrowed the same way (rbs_set_observation_borrowed_f32)
        a.set_observation(frames[1][1]); b.set_observation_borrowed(frames[1][1].astype(np.float32))
        la, lb = a.loglikes_poses(poses[1], ia, update=True), b.loglikes_poses(poses[1], ib, update=True)
        assert np.array_equal(la, lb)
        # staged by whatever needs the observation first
        f64 = frames[0][1].astype(np.float64)
        b.set_observation_borrowed(f64)
        assert np.array_equal(b.get_observation(), frames[0][1], equal_nan=True)
        b.set_observation_borrowed(f64)
        b.synchronize()
        a.set_observation(f64); a.set_observation(f64)
        d_poses = torch.from_numpy(poses[0].reshape(n, -1)).cuda()
        d_idx = torch.from_numpy(parents[0]).cuda()
        outs = []
        for s_ in (a, b):
            if s_ is b:
                b.set_observation_borrowed(f64)
            else:
                a.set_observation(f64)
            d_out = torch.empty(n, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            s_.loglikes_device(d_poses.data_ptr(), d_idx.data_ptr(), n, False, d_out.data_ptr())
            s_.synchronize()
            outs.append(d_out.cpu().numpy())
        assert np.array_equal(outs[0], outs[1])
        # a likelihood call that is REFUSED gives the frame back all the same: it is copied on the way out (the observation it
        # would have been), and what the caller does to the buffer afterwards is the caller's business
        g64 = frames[2][1].astype(np.float64)
        a.set_observation(g64)
        mine = g64.copy()
        b.set_observation_borrowed(mine)
        bad = parents[2].copy(); bad[3] = n + 5
        with pytest.raises(Exception):
            b.loglikes_poses(poses[2], bad, update=True)
        mine[:] = 0.123
        ia, ib = parents[1].copy(), parents[1].copy()
        la, lb = a.loglikes_poses(poses[2], ia, update=True), b.loglikes_poses(poses[2], ib, update=True)
        assert np.array_equal(la, lb)


@pytest.mark.parametrize("meshes,cols,rows,n,slab", [(("m1",), 640, 480, 128, 0), (("m1_l2", "box12"), 320, 240, 96, 0), (("m1",), 640, 480, 96, 16384),
                                                       (("m4",), 640, 480, 32, 0), (("m4",), 640, 480, 32, 16384)])   # (m4: the many-cluster kernels)
@pytest.mark.parametrize("precision", ["f64", "f32"])      # (f32: round 6 -- the float32 likelihood's kernels take the shared plane too)
def test_shared_trail_stores_the_same_planes(gpu_lib, monkeypatch, meshes, cols, rows, n, slab, precision):
    """The shared background plane (rbsensor_mi355x.h "shared trail") changes what is STORED, not a bit of what is computed: a
    handle forced into it (RBS_STP_ENTER=0: at the first sampled window area; re-based every 3rd updating call) against a handle
    that never uses it, on a tracked sequence whose object travels across the image with resampling (children share parents):
    log-likelihoods and whole planes bit for bit, read-only calls included -- and the windows really are smaller."""
    nb = len(meshes)
    om, cam, P = sc.make_scene(meshes, cols, rows, max_particles=n)
    o = ob.Oracle(om, cam, P, max_particles=n, mode=ob.EAGER)
    rng = np.random.default_rng(31)
    frames = []
    for k in range(26):
        t = synth.truth_pose(nb, frame=k)
        t[:, 9] += -0.12 + 0.01 * k            # 1 cm per frame across the image: a long trail
        t[:, 10] += -0.06 + 0.005 * k
        frames.append((t, synth.make_frame(o.render_depth(t), rows, cols, rng)))
    poses = [synth.particle_poses(t, n, rng, scale=1.0) for t, _ in frames]
    parents = [np.sort(rng.choice(n, size=n, p=(lambda w: w / w.sum())(rng.random(n) ** 8))).astype(np.int32) for _ in frames]   # few survivors
    monkeypatch.setenv("RBS_SHARED_TRAIL", "0")
    with RbSensor(om, cam, P, max_particles=n, precision=precision) as plain:      # (whole planes, scalar background: the reference run)
        monkeypatch.setenv("RBS_SHARED_TRAIL", "1")
        monkeypatch.setenv("RBS_STP_ENTER", "0.0")
        monkeypatch.setenv("RBS_STP_EVERY", "3")
        with RbSensor(om, cam, P, max_particles=n, precision=precision, slab_px=slab) as g:      # (slab > 0: window-sized slabs, which grow on the way)
            g.set_timing_every(1); plain.set_timing_every(1)      # (the window area is sampled on timed calls)
            for s_ in (g, plain):
                s_.reset()
            o.reset()
            ig, ip, io = (np.zeros(n, np.int32) for _ in range(3))
            for k, (_, frame) in enumerate(frames):
                for s_ in (g, plain, o):
                    s_.set_observation(frame)
                if k % 5 == 4:      # a read-only block in front of the updating one
                    ra, rb = g.loglikes_poses(poses[k - 1], ig.copy(), update=False), plain.loglikes_poses(poses[k - 1], ip.copy(), update=False)
                    assert np.array_equal(ra, rb), (k, np.abs(ra - rb).max())
                la, lb = g.loglikes_poses(poses[k], ig, update=True), plain.loglikes_poses(poses[k], ip, update=True)
                lo = o.loglikes_poses(poses[k], io, update=True)
                assert np.array_equal(la, lb), (k, np.abs(la - lb).max())
                assert rel_err(la, lo).max() <= (TOL_EAGER if precision == "f64" else 1e-3)   # (f32: float32-level agreement, tests/test_gpu_f32.py)
                ig, ip, io = parents[k].copy(), parents[k].copy(), parents[k].copy()
            active, rebases = g.shared_trail_state()
            assert active and rebases >= 3, (active, rebases)
            assert plain.shared_trail_state() == (False, 0)
            area = lambda w: max(0, w[2] - w[0]) * max(0, w[3] - w[1])
            slots = list(range(0, n, max(1, n // 16)))
            a_g, a_p = np.mean([area(g.get_window(q)) for q in slots]), np.mean([area(plain.get_window(q)) for q in slots])
            print(f"\nmean window: shared trail {a_g:.0f} px, scalar background {a_p:.0f} px ({rebases} re-basings)")
            assert a_g < 0.75 * a_p
            if slab:     # window transport on slabs against a shared plane: the plane travels whole, and comes back the same
                import torch
                buf = torch.empty(cols * rows, dtype=torch.float32, device="cuda:0")
                rect = g.export_window(slots[1], buf.data_ptr(), buf.numel())
                assert rect == (0, 0, cols, rows)
                g.import_window(n - 1, rect, buf.data_ptr())
                g.synchronize()
                assert np.array_equal(g.get_occlusion(n - 1), g.get_occlusion(slots[1]))
            for q in slots:
                assert np.array_equal(g.get_occlusion(q), plain.get_occlusion(q)), q
                if precision == "f64":
                    assert np.array_equal(g.get_occlusion(q), o.get_occlusion(q)) or (g.get_occlusion(q) != o.get_occlusion(q)).mean() <= 1e-4


@pytest.mark.parametrize("slab", [0, 2048])
def test_shared_trail_is_left_when_the_particles_share_nothing(gpu_lib, monkeypatch, slab):
    """Particles that never resample (every child its own parent's only child) share no ancestor: re-basing the shared plane
    on one of them shrinks nobody else's window.  The handle notices -- windows still most of the frame 16 calls after a
    re-basing -- and goes back to the scalar background (and the whole-plane machinery such windows are served best by).
    Bit-identical planes and log-likelihoods all the way, through entering, re-basing and leaving."""
    n, cols, rows = 24, 160, 120
    om, cam, P = sc.make_scene(("m1_l2",), cols, rows, max_particles=n)
    o = ob.Oracle(om, cam, P, max_particles=1, mode=ob.EAGER)
    rng = np.random.default_rng(17)
    frames = []
    for k in range(110):
        t = synth.truth_pose(1, frame=0)
        s_ = min(k, 70) / 70.0
        t[0, 9], t[0, 10] = -0.32 + 0.64 * s_, -0.23 + 0.46 * s_       # corner to corner in 70 frames, then rest: a window of most of the frame
        frames.append((t, synth.make_frame(o.render_depth(t), rows, cols, rng)))
    poses = [synth.particle_poses(t, n, rng, scale=1.0) for t, _ in frames]
    monkeypatch.setenv("RBS_SHARED_TRAIL", "0")
    with RbSensor(om, cam, P, max_particles=n, precision="f64") as plain:
        monkeypatch.setenv("RBS_SHARED_TRAIL", "1")
        monkeypatch.setenv("RBS_STP_ENTER", "0.05")
        with RbSensor(om, cam, P, max_particles=n, precision="f64", slab_px=slab) as g:      # (slabs: the call that leaves re-measures every
            g.set_timing_every(1); plain.set_timing_every(1)                                # child against the scalar level -- regions that no
            g.reset(); plain.reset()                                                        # longer fit, a call taken back and repeated)
            ig, ip = np.arange(n, dtype=np.int32), np.arange(n, dtype=np.int32)
            states = []
            for k, (_, frame) in enumerate(frames):
                g.set_observation(frame); plain.set_observation(frame)
                la, lb = g.loglikes_poses(poses[k], ig, update=True), plain.loglikes_poses(poses[k], ip, update=True)
                assert np.array_equal(la, lb), (k, np.abs(la - lb).max())
                states.append(g.shared_trail_state())
            entered = next(k for k, st in enumerate(states) if st[0])
            left = next(k for k in range(entered, len(states)) if not states[k][0])
            print(f"\nshared trail entered at call {entered}, re-based {states[left][1]}x, left at call {left}")
            assert states[-1][0] is False and states[-1][1] >= 1
            for q in range(0, n, 5):
                assert np.array_equal(g.get_occlusion(q), plain.get_occlusion(q)), q


def test_shared_trail_survives_the_first_borrowed_frame(gpu_lib, monkeypatch):
    """The hand-over buffer of the two-kernel launch is allocated at the first call that needs it -- here a borrowed frame arriving
    AFTER the handle entered the shared trail on one-kernel launches.  (Round 5: that allocation used to free the shared planes
    it had nothing to do with and went on using them -- harmless only as long as nobody else was handed that memory, which is
    why this sequence passed then too; it is kept as the sequence, with foreign allocations of a plane's size in between.)  Same
    bits as a handle that never shares, before and after."""
    n, cols, rows = 48, 320, 240
    om, cam, P = sc.make_scene(("m1_l2",), cols, rows, max_particles=n)
    o = ob.Oracle(om, cam, P, max_particles=1, mode=ob.EAGER)
    rng = np.random.default_rng(5)
    frames = []
    for k in range(16):
        t = synth.truth_pose(1, frame=k)
        t[:, 9] += -0.08 + 0.01 * k
        frames.append((t, synth.make_frame(o.render_depth(t), rows, cols, rng)))
    poses = [synth.particle_poses(t, n, rng, scale=1.0) for t, _ in frames]
    parents = [np.sort(rng.choice(n, size=n, p=(lambda w: w / w.sum())(rng.random(n) ** 8))).astype(np.int32) for _ in frames]
    monkeypatch.setenv("RBS_SPLIT", "0")
    monkeypatch.setenv("RBS_SHARED_TRAIL", "0")
    with RbSensor(om, cam, P, max_particles=n, precision="f64") as plain:
        monkeypatch.setenv("RBS_SHARED_TRAIL", "1")
        monkeypatch.setenv("RBS_STP_ENTER", "0.0")
        monkeypatch.setenv("RBS_STP_EVERY", "3")
        with RbSensor(om, cam, P, max_particles=n, precision="f64") as g:
            g.set_timing_every(1); plain.set_timing_every(1)
            g.reset(); plain.reset()
            ig, ip = np.zeros(n, np.int32), np.zeros(n, np.int32)
            for k, (_, frame) in enumerate(frames):
                plain.set_observation(frame)
                if k >= 8:
                    assert g.shared_trail_state()[0]
                    g.set_observation_borrowed(np.ascontiguousarray(frame, dtype=np.float64))   # two-kernel launch: its buffer appears at k == 8
                else:
                    g.set_observation(frame)
                la, lb = g.loglikes_poses(poses[k], ig, update=True), plain.loglikes_poses(poses[k], ip, update=True)
                assert np.array_equal(la, lb), (k, np.abs(la - lb).max())
                ig, ip = parents[k].copy(), parents[k].copy()
                if k == 8:      # somebody else's allocations of a plane's size, written: freed planes would be handed out again here
                    import torch
                    junk = [torch.full((cols * rows,), float("nan"), dtype=torch.float32, device="cuda:0") for _ in range(8)]
                    with RbSensor(om, cam, P, max_particles=n, precision="f64") as other:
                        other.reset()
                        other.set_observation(np.full_like(frame, 0.3))
                        other.loglikes_poses(poses[k], np.zeros(n, np.int32), update=True)
                    torch.cuda.synchronize()
            del junk
            for q in range(0, n, 6):
                assert np.array_equal(g.get_occlusion(q), plain.get_occlusion(q)), q


@pytest.mark.parametrize("n", [300, 9000])
def test_tracker_takes_double_frames_and_stages_them_behind_the_geometry_kernel(gpu_lib, monkeypatch, n):
    """rbs_tracker_track_f64 / _submit_f64 (the image as dbot's tracker receives it: doubles) against the float entry points, frame by
    frame and with look-ahead; n = 300: the frame is staged between the two kernels of the split launch (RBS_TRACKER_SPLIT_MAX), n = 9 000:
    it is copied first and the one-kernel launch runs -- the estimates are the same bits every way (device RNG, same seed)."""
    from dbot_ros_amd import pose
    from dbot_ros_amd.tracker import DeviceParticleTracker, ObjectTransitionBuilder, ParticleTrackerBuilder
    cols, rows = (320, 240)
    om, cam, P = sc.make_scene(("m1_l2",), cols, rows, max_particles=n)
    rng = np.random.default_rng(3)
    init = np.zeros(12)
    Rt = synth.truth_pose(1, frame=0)[0]
    init[3:6] = pose.matrix_to_rotvec(Rt[:9].reshape(3, 3))
    init[0:3] = Rt[9:] - Rt[:9].reshape(3, 3) @ om.centers[0]
    runs = {}
    for tag, env, dtype, ahead in (("f32", "5000", np.float32, False), ("f64", "5000", np.float64, False), ("f64_mono", "0", np.float64, False),
                                   ("f64_ahead", "5000", np.float64, True)):
        monkeypatch.setenv("RBS_TRACKER_SPLIT_MAX", env)
        with RbSensor(om, cam, P, max_particles=n, precision="f64") as s:
            if tag == "f32":
                frames = [synth.make_frame(s.render_depth(synth.truth_pose(1, frame=k)), rows, cols, rng, occluder=False) for k in range(8)]
            trans = ObjectTransitionBuilder(ObjectTransitionBuilder.Parameters()).build()
            tr = DeviceParticleTracker(trans, s, om, ParticleTrackerBuilder.Parameters(evaluation_count=n), device_rng=True, seed=9)
            tr.initialize([init])
            ests = []
            if ahead:
                tr.submit(frames[0].astype(dtype))
                for k in range(1, 8):
                    tr.submit(frames[k].astype(dtype))
                    ests.append(tr.result())
                ests.append(tr.result())
            else:
                ests = [tr.track(f.astype(dtype)) for f in frames]
            runs[tag] = np.stack(ests)
            tr.close()
    for tag in ("f64", "f64_mono", "f64_ahead"):
        assert np.array_equal(runs[tag], runs["f32"]), (tag, np.abs(runs[tag] - runs["f32"]).max())
