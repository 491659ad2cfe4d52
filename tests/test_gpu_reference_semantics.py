"""Reference-semantics parity at BASELINE sizes: the HIP path against the LAZY oracle.

The reference CPU model (use_gpu:false, R:source/dbot_ros/tracker/particle_tracker_node.cpp:165) keeps
a per-pixel time stamp and propagates a pixel's occlusion by its own elapsed time in double
(constants R:...particle_tracker_node.cpp:176-189, delta_time :189) -- the oracle's
LAZY mode.  The device advances every stored value on every updating call with one float FMA and
snaps values within 2^-18 of the never-covered level onto it (the oracle's EAGER mode, DESIGN ledger
L8 / L9).  The other full-size tests compare with the EAGER oracle (to 1e-9); the tests here hold the
device to north_star's bar against the LAZY one, at the sizes BASELINE.json names:

  * every particle of C1 (2 000, 640x480, M1) on every frame of a 30-frame tracked sequence with
    KL-triggered multinomial resampling:  |d ll| <= 1e-5 max(1, |ll|);
  * the PARENT INDICES each resampling of that sequence draws from the device's log-likelihoods
    against those it draws from the LAZY oracle's, same uniforms, same history (after counting, both
    sides continue with the oracle's parents, so every resampling is compared on identical inputs):
    the number of children whose parent differs is printed and bounded, and every such child must be
    a uniform within 1e-7 of a step of the cumulative weights (it drew the neighbouring parent, parents
    of weight ~0 aside);
  * the same at 20 000 particles (the reference's default 80x60 operating point, where the CPU
    oracle can follow 20 000 particles);
  * the device's occlusion planes after the 30 frames against orc_get_occlusion_now (LAZY);
  * C1 (all 2 000) and C2 (a random 1 000 of 6 666 x 3 bodies) in the two-frame full-size harness of
    test_gpu_fullsize.py with the LAZY oracle as the checker; C4's geometry (M4 at 1280x960, 256 particles) likewise;
  * C2's structure as a 30-frame sequence: three bodies, three sampling blocks per frame, resampling after any block;
  * rbs_tracker_* at 20 000 particles against oracle/tracker_oracle.c over the LAZY sensor.
"""
import numpy as np
import pytest

import oracle_binding as ob
import scenarios as sc
from dbot_ros_amd import RbSensor, filter as flt, pose, synth
from dbot_ros_amd.pose import pack_Rt, rotvec_to_matrix

pytestmark = pytest.mark.gpu

TOL_LAZY = 1e-5          # north_star: "within 1e-5 float tolerance"
CDF_DELTA = 1e-7         # a child whose parent differs drew a uniform this close to a cdf step
PLANE_TOL = 6e-6         # |device plane - LAZY plane "as of now"|: the snap (2^-18 = 3.8e-6, once) + float steps


def rel_err(a, b):
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


def _poses_around(truth, dl, da):
    """Particles at R_b = R(da_b) R_truth_b, t_b = t_truth_b + dl_b for every body b (dl, da: [n, 3] for one body or [n, bodies, 3])."""
    truth = np.asarray(truth).reshape(-1, 12)
    nb = truth.shape[0]
    dl = np.asarray(dl).reshape(len(dl), nb, 3)
    da = np.asarray(da).reshape(len(da), nb, 3)
    R0 = truth[:, :9].reshape(1, nb, 3, 3)
    return pack_Rt(rotvec_to_matrix(da) @ R0, truth[None, :, 9:12] + dl)


def _lockstep(meshes, cols, rows, n, n_frames, seed, plane_slots, occlusion=None, tol=TOL_LAZY, truth_fn=None, info=None, **sensor_kw):
    """The filter step (log-weights, KL test, multinomial resampling from host-supplied uniforms:
    dbot_ros_amd/filter.py) driven twice on identical inputs: once from the device's
    log-likelihoods, once from the LAZY oracle's.  Particles are a random walk around the moving
    ground truth and are inherited by the children.  Several bodies: one SAMPLING BLOCK per body and frame
     -- block b moves body b only, the blocks before the last are read-only calls, the last one
    updates; weights, KL test and resampling after every block, the parent indices carried from block to
    block.  Returns per-resampling mismatch counts, the worst log-likelihood error and the worst plane difference."""
    nb = len(meshes)
    om, cam, P = sc.make_scene(meshes, cols, rows, max_particles=n)
    lazy = ob.Oracle(om, cam, P, max_particles=n, mode=ob.LAZY)
    threads = sc.usable_threads()
    rng = np.random.default_rng(seed)
    dl = rng.normal(0.0, 0.0025, size=(n, nb, 3))
    da = rng.normal(0.0, 0.02, size=(n, nb, 3))
    idx_g, idx_o = np.zeros(n, np.int32), np.zeros(n, np.int32)
    logw = [np.zeros(n), np.zeros(n)]     # device, oracle
    ll_prev = [np.zeros(n), np.zeros(n)]
    mismatches, worst_ll, n_children = [], 0.0, 0
    with RbSensor(om, cam, P, max_particles=n, occlusion=occlusion, **sensor_kw) as g:
        g.reset()
        lazy.reset(threads=threads)
        for k in range(n_frames):
            truth = truth_fn(k) if truth_fn else synth.truth_pose(nb, frame=k)
            frame = synth.make_frame(lazy.render_depth(truth), rows, cols, rng)
            g.set_observation(frame)
            lazy.set_observation(frame)
            for blk in range(nb):
                # block blk's transition: body blk moves (AR(1) around the truth), the bodies of later blocks keep last frame's delta
                dl[:, blk] = 0.8 * dl[:, blk] + rng.normal(0.0, 0.0025, size=(n, 3))
                da[:, blk] = 0.8 * da[:, blk] + rng.normal(0.0, 0.02, size=(n, 3))
                poses = _poses_around(truth, dl, da)
                upd = blk == nb - 1
                ll = [g.loglikes_poses(poses, idx_g, update=upd),
                      lazy.loglikes_poses(poses, idx_o, update=upd, threads=threads)]
                e = rel_err(ll[0], ll[1])
                assert e.max() <= tol, (k, blk, int(e.argmax()), float(e.max()))      # EVERY particle
                worst_ll = max(worst_ll, float(e.max()))
                w, kl = [], []
                for s in range(2):
                    logw[s] += ll[s] - ll_prev[s]
                    ll_prev[s] = ll[s]
                    w.append(flt.normalized_weights(logw[s]))
                    kl.append(flt.kl_to_uniform(w[s]))
                assert (kl[0] > 2.0) == (kl[1] > 2.0), (k, blk, kl)
                if kl[1] > 2.0:
                    u = rng.random(n)
                    pg, po = flt.multinomial_resample(w[0], u), flt.multinomial_resample(w[1], u)
                    bad = np.nonzero(pg != po)[0]
                    mismatches.append(len(bad))
                    n_children += n
                    if len(bad):
                        # each one drew a uniform within CDF_DELTA of the cumulative weights of EVERY parent between
                        # the two answers (neighbours, or neighbours but for parents of weight ~0 between them)
                        c = np.cumsum(w[1])
                        c /= c[-1]
                        a, b = np.minimum(pg[bad], po[bad]), np.maximum(pg[bad], po[bad])
                        worst = np.maximum(np.abs(u[bad] - c[a]), np.abs(u[bad] - c[b - 1]))
                        assert (worst <= CDF_DELTA).all(), (k, pg[bad], po[bad], worst)
                    # both sides continue with the REFERENCE-semantics parents: identical histories
                    dl, da = dl[po], da[po]
                    # (the occlusion slot a child inherits: its parent's -- after an updating call that is the parent's own
                    # index, after a read-only one whatever slot the parent itself was still pointing at)
                    idx_g, idx_o = idx_g[po].copy(), idx_o[po].copy()
                    for s in range(2):
                        ll_prev[s] = ll_prev[s][po]
                        logw[s] = np.zeros(n)
        # the planes the next frame would start from: device (stored, eager) against LAZY "as of now"
        worst_plane, n_diff = 0.0, 0
        for slot in rng.choice(n, size=min(plane_slots, n), replace=False):
            d = np.abs(g.get_occlusion(int(slot)) - lazy.get_occlusion(int(slot), now=True))
            worst_plane = max(worst_plane, float(d.max()))
            n_diff += int((d > 1e-6).sum())
        if info is not None:      # what the handle stored at the end: the shared trail's state, the mean stored window
            try:
                info["shared_trail"] = g.shared_trail_state()
            except Exception as e:      # (a handle over several devices answers per shard)
                info["shared_trail"] = repr(e)
            area = lambda w: max(0, w[2] - w[0]) * max(0, w[3] - w[1])
            info["window_fraction"] = float(np.mean([area(g.get_window(int(q))) for q in range(0, n, max(1, n // 32))])) / (rows * cols)
    lazy.close()
    return mismatches, n_children, worst_ll, worst_plane, n_diff


def test_c1_sequence_every_particle_and_parents_vs_lazy_oracle(gpu_lib):
    """BASELINE C1 (2 000 particles, M1, 640x480), 30 frames."""
    mism, children, worst_ll, worst_plane, n_diff = _lockstep(("m1",), 640, 480, 2000, 30, seed=41, plane_slots=200)
    print(f"\nC1 vs LAZY oracle: worst |d ll| / max(1,|ll|) = {worst_ll:.3e}; {len(mism)} resamplings, "
          f"parent mismatches per resampling {mism} = {sum(mism)} of {children} children; "
          f"planes: worst |d| = {worst_plane:.3e}, {n_diff} pixels above 1e-6 in 200 planes")
    assert len(mism) >= 5
    # measured (round 4): 0 of 60 000 children over 30 resamplings
    assert max(mism) <= 2 and sum(mism) <= max(2, children // 10000), mism
    assert worst_plane <= PLANE_TOL, worst_plane


# ---- rbs_config.occlusion_mode = REFERENCE (round 6): the reference's own bookkeeping on the device ----
TOL_EXACT = 1e-10        # what is left is rbs_math's exp / erfc / log against libm's: measured ~1e-13
PLANE_EXACT = 1.2e-7     # planes: bit for bit, but for a posterior whose float rounding sits within 1e-16 of a boundary (one ulp)


def test_c1_sequence_reference_mode_parents_identical(gpu_lib):
    """BASELINE C1, 30 frames, occlusion_mode REFERENCE: every particle at the transcendentals' accuracy, EVERY resampling's
    parents identical to the LAZY oracle's, planes bit for bit."""
    mism, children, worst_ll, worst_plane, n_diff = _lockstep(("m1",), 640, 480, 2000, 30, seed=41, plane_slots=200, occlusion="reference", tol=TOL_EXACT)
    print(f"\nC1, occlusion_mode REFERENCE vs LAZY oracle: worst |d ll| / max(1,|ll|) = {worst_ll:.3e}; {len(mism)} resamplings, "
          f"{sum(mism)} of {children} children with another parent; planes: worst |d| = {worst_plane:.3e}")
    assert len(mism) >= 5 and sum(mism) == 0, mism
    assert worst_plane <= PLANE_EXACT, worst_plane


def test_20000_particles_reference_mode_parents_identical(gpu_lib):
    """20 000 particles at 80x60, 30 resamplings = 600 000 children: none draws another parent (the device rule: 16)."""
    mism, children, worst_ll, worst_plane, n_diff = _lockstep(("m1_l2",), 80, 60, 20000, 30, seed=43, plane_slots=500, occlusion="reference", tol=TOL_EXACT)
    print(f"\n20 000 particles, occlusion_mode REFERENCE vs LAZY oracle: worst |d ll| / max(1,|ll|) = {worst_ll:.3e}; {len(mism)} resamplings, "
          f"{sum(mism)} of {children} children with another parent; planes: worst |d| = {worst_plane:.3e}")
    assert len(mism) >= 5 and sum(mism) == 0, mism
    assert worst_plane <= PLANE_EXACT, worst_plane


def test_c2_sequence_reference_mode_parents_identical(gpu_lib):
    """C2's structure (three bodies, two read-only blocks and the updating one per frame, resampling after any block)."""
    mism, children, worst_ll, worst_plane, n_diff = _lockstep(("m1", "m2", "m3"), 640, 480, 600, 20, seed=47, plane_slots=40, occlusion="reference", tol=TOL_EXACT)
    print(f"\nC2 structure, occlusion_mode REFERENCE vs LAZY oracle: worst |d ll| / max(1,|ll|) = {worst_ll:.3e}; {len(mism)} resamplings, "
          f"{sum(mism)} of {children} children with another parent; planes: worst |d| = {worst_plane:.3e}")
    assert len(mism) >= 5 and sum(mism) == 0, mism
    assert worst_plane <= PLANE_EXACT, worst_plane


def test_20000_particles_parents_vs_lazy_oracle(gpu_lib):
    """20 000 particles at the reference's default 80x60 (R:config/camera.yaml: downsampling 8), 30 frames."""
    mism, children, worst_ll, worst_plane, n_diff = _lockstep(("m1_l2",), 80, 60, 20000, 30, seed=43, plane_slots=500)
    print(f"\n20 000 particles vs LAZY oracle: worst |d ll| / max(1,|ll|) = {worst_ll:.3e}; {len(mism)} resamplings, "
          f"parent mismatches per resampling {mism} = {sum(mism)} of {children} children; "
          f"planes: worst |d| = {worst_plane:.3e}, {n_diff} pixels above 1e-6 in 500 planes")
    assert len(mism) >= 5
    # measured (round 4): 16 of 600 000 children over 30 resamplings, at most 5 in one
    assert max(mism) <= 12 and sum(mism) <= children // 10000, mism
    assert worst_plane <= PLANE_TOL, worst_plane


def _two_frames(meshes, cols, rows, n, k_oracle, blocks_readonly, chunk, z=0.7, occlusion=None, tol=None):
    """test_gpu_fullsize._full_size_case's two-frame run with the LAZY oracle as the checker."""
    nb = len(meshes)
    om, cam, P = sc.make_scene(meshes, cols, rows, max_particles=n)
    render = ob.Oracle(om, cam, P, max_particles=1, mode=ob.LAZY)
    rng = np.random.default_rng(21)
    truth = synth.truth_pose(nb, z=z)
    frame = synth.make_frame(render.render_depth(truth), rows, cols, rng)
    render.close()
    poses = synth.particle_poses(truth, n, rng, scale=2.0)
    with RbSensor(om, cam, P, max_particles=n, occlusion=occlusion) as g:
        g.reset()
        g.set_observation(frame)
        idx = np.zeros(n, np.int32)
        out = [g.loglikes_poses(poses, idx.copy(), update=False) for _ in range(blocks_readonly)]
        out.append(g.loglikes_poses(poses, idx, update=True))
        g.set_observation(frame)
        out.append(g.loglikes_poses(poses, np.arange(n, dtype=np.int32), update=True))
    sel = np.arange(n) if k_oracle >= n else np.sort(rng.choice(n, size=k_oracle, replace=False))
    threads = sc.usable_threads()
    worst = 0.0
    for lo in range(0, len(sel), chunk):
        part = sel[lo:lo + chunk]
        k = len(part)
        orc = ob.Oracle(om, cam, P, max_particles=k, mode=ob.LAZY)
        orc.reset(threads=threads)
        orc.set_observation(frame)
        io = np.zeros(k, np.int32)
        r = [orc.loglikes_poses(poses[part], io.copy(), update=False, threads=threads) for _ in range(blocks_readonly)]
        r.append(orc.loglikes_poses(poses[part], io, update=True, threads=threads))
        orc.set_observation(frame)
        r.append(orc.loglikes_poses(poses[part], np.arange(k, dtype=np.int32), update=True, threads=threads))
        orc.close()
        for x, y in zip(out, r):
            e = rel_err(x[part], y)
            assert e.max() <= (TOL_LAZY if tol is None else tol), float(e.max())
            worst = max(worst, float(e.max()))
    return worst


def test_c1_full_size_every_particle_vs_lazy_oracle(gpu_lib):
    worst = _two_frames(("m1",), 640, 480, 2000, 2000, 0, 500)
    print(f"\nC1, all 2 000 particles vs LAZY oracle: {worst:.3e}")


@pytest.mark.parametrize("occlusion", ["device", "reference"])
def test_c2_full_size_random_1000_vs_lazy_oracle(gpu_lib, occlusion):
    """BASELINE C2: 6 666 particles x meshes [M1, M2, M3] (R:config/object.yaml:3-5 lists several meshes),
    two read-only blocks and the updating one per frame; a random 1 000 against the LAZY oracle."""
    worst = _two_frames(("m1", "m2", "m3"), 640, 480, 6666, 1000, 2, 500, occlusion=occlusion, tol=TOL_EXACT if occlusion == "reference" else None)
    print(f"\nC2, a random 1 000 particles vs LAZY oracle, occlusion {occlusion}: {worst:.3e}")


@pytest.mark.parametrize("occlusion", ["device", "reference"])
def test_c4_geometry_vs_lazy_oracle(gpu_lib, occlusion):
    """BASELINE C4's geometry -- M4 (50 880 triangles) at 1280x960, the object at 0.5 m (8 work items per particle,
    the shared cluster cull of the many-cluster kernels) -- 256 particles, two frames, against the LAZY oracle (VERDICT r4 #4a);
    occlusion reference: the stamped-plane form of those kernels (rbs_raster_kernel_exact_f64<.., MANY = true, ..>) at its own bar."""
    worst = _two_frames(("m4",), 1280, 960, 256, 256, 0, 128, z=0.5, occlusion=occlusion, tol=TOL_EXACT if occlusion == "reference" else None)
    print(f"\nC4 geometry (M4, 1280x960), 256 particles vs LAZY oracle, occlusion {occlusion}: {worst:.3e}")


def test_c2_sequence_three_bodies_three_blocks_vs_lazy_oracle(gpu_lib):
    """BASELINE C2's structure as a tracked sequence (VERDICT r4 #4b): meshes [M1, M2, M3] (R:config/object.yaml:3-5 lists several),
    three sampling blocks per frame -- two read-only calls and the updating one -- 30 frames, KL-triggered multinomial resampling
    after every block, EVERY particle on every block against the LAZY oracle, the parents of every resampling compared.
    1 200 particles (the CPU oracle renders 1 200 x 3 bodies x 90 calls)."""
    mism, children, worst_ll, worst_plane, n_diff = _lockstep(("m1", "m2", "m3"), 640, 480, 1200, 30, seed=47, plane_slots=60)
    print(f"\nC2 sequence (3 bodies, 3 blocks per frame, 1 200 particles) vs LAZY oracle: worst |d ll| / max(1,|ll|) = {worst_ll:.3e}; "
          f"{len(mism)} resamplings, parent mismatches per resampling {mism} = {sum(mism)} of {children} children; "
          f"planes: worst |d| = {worst_plane:.3e}, {n_diff} pixels above 1e-6 in 60 planes")
    assert len(mism) >= 5
    assert max(mism) <= 2 and sum(mism) <= max(2, children // 10000), mism
    assert worst_plane <= PLANE_TOL, worst_plane


@pytest.mark.parametrize("occlusion", ["device", "reference"])
def test_device_tracker_20000_particles_vs_lazy_oracle_tracker(gpu_lib, occlusion):
    """rbs_tracker_* (default precision F64) at 20 000 particles against oracle/tracker_oracle.c over
    the LAZY (reference-semantics) sensor, same host-supplied randomness.  As long as every earlier
    resampling drew identical parents the two runs have identical histories and the parents of the
    next one are compared child by child (count printed and bounded); after a first difference -- one
    child with the neighbouring parent shifts every later cumulative weight by ~1/n, so indices stop
    being comparable -- the estimates are only held to the spread of the particle cloud (2e-3: the two
    runs are then two draws of the same filter); before it, to 1e-6."""
    from dbot_ros_amd.tracker import DeviceParticleTracker, ObjectTransitionBuilder, ParticleTrackerBuilder
    n, cols, rows = 20000, 80, 60
    om, cam, P = sc.make_scene(("m1_l2",), cols, rows, max_particles=n)
    tp = ParticleTrackerBuilder.Parameters(evaluation_count=n, center_object_frame=False)
    orc = ob.Oracle(om, cam, P, max_particles=n, mode=ob.LAZY)
    trans = ObjectTransitionBuilder(ObjectTransitionBuilder.Parameters()).build()
    ref = ob.OracleTracker(orc, n, trans.sigma, trans.vf, tp.max_kl_divergence)
    init = np.zeros(12)
    Rt = synth.truth_pose(1, frame=0)[0]
    init[3:6] = pose.matrix_to_rotvec(Rt[:9].reshape(3, 3))
    init[0:3] = Rt[9:]
    counts, same_history, worst = [], True, [0.0, 0.0]
    with RbSensor(om, cam, P, max_particles=n, occlusion=occlusion) as s:
        dev = DeviceParticleTracker(trans, s, om, tp, np.random.default_rng(2))
        dev.initialize([init])
        ref.initialize(init)
        rng = np.random.default_rng(10)
        nres_prev = 0
        for k in range(1, 9):
            frame = synth.make_frame(orc.render_depth(synth.truth_pose(1, frame=k)), rows, cols, rng, occluder=False)
            normals, uniforms = dev.draw_randomness()
            ed = dev.track(frame, normals, uniforms)
            er, nres = ref.track(frame, normals, uniforms)
            assert dev.n_resamplings == nres, (k, dev.n_resamplings, nres)
            err = float(np.abs(ed - er).max())
            if same_history and nres > nres_prev:
                _, _, idd = dev.get_state()
                _, _, idr = ref.get_state()
                bad = int((idd != idr).sum())
                counts.append(bad)
                same_history = bad == 0
            worst[0 if same_history else 1] = max(worst[0 if same_history else 1], err)
            assert err <= (1e-6 if same_history else 2e-3), (k, same_history, err)
            nres_prev = nres
        dev.close()
    print(f"\ndevice tracker vs LAZY oracle tracker, 20 000 particles: parent mismatches per resampling "
          f"(identical histories up to the first difference) {counts}; estimates differ by <= {worst[0]:.2e} "
          f"up to it, <= {worst[1]:.2e} after")
    assert len(counts) >= 1 and max(counts) <= (0 if occlusion == "reference" else 12), counts


# ---- the moving-object regime against the LAZY oracle ----
def _sweep_truth(cols):
    """The object crosses more than half the image and returns over its own trail: 100 frames out, 100 back."""
    def truth(k):
        t = synth.truth_pose(1, frame=0)
        phase = k if k <= 100 else 200 - k
        t[0, 9] += -0.25 + 0.005 * phase          # 0.5 m at z = 0.7: ~0.64 of the image width
        t[0, 10] += 0.04 * np.sin(0.05 * k)
        return t
    return truth


@pytest.mark.parametrize("layout", ["planes", "slabs", "two_shards"])
@pytest.mark.parametrize("occlusion", ["reference", "device"])
def test_moving_object_over_its_own_trail_vs_lazy_oracle(gpu_lib, occlusion, layout):
    """An object that sweeps 0.64 of the image width and comes back over its own trail, 200 frames at 320x240, 256 particles,
    KL-triggered multinomial resampling, against the LAZY (reference-semantics) oracle: every particle on every frame, the
    parents of every resampling, the planes the next frame would start from.  This is the regime where the stored windows
    grow and the handle switches to the shared trail BY ITS OWN THRESHOLD (no RBS_STP_* in the environment; the state is
    printed), and where eager-versus-lazy drift on the trail's pixels is largest.  Whole planes, window-sized slabs, and one
    handle over two shards (a shard keeps the scalar background).  occlusion "reference": the transcendentals' accuracy and
    not one parent different; "device": north_star's 1e-5 and the float-stepped rule's usual handful of neighbouring parents."""
    cols, rows, n = 320, 240, 256
    kw = {"planes": {}, "slabs": {"slab_px": 8192}, "two_shards": {"device_ids": [0, 0]}}[layout]
    info = {}
    exact = occlusion == "reference"
    mism, children, worst_ll, worst_plane, n_diff = _lockstep(("m1_l2",), cols, rows, n, 200, seed=53, plane_slots=32, occlusion=occlusion,
                                                              tol=TOL_EXACT if exact else TOL_LAZY, truth_fn=_sweep_truth(cols), info=info, **kw)
    print(f"\nmoving object, {layout}, occlusion {occlusion}: worst |d ll| / max(1,|ll|) = {worst_ll:.3e}; {len(mism)} resamplings, {sum(mism)} of {children} "
          f"children with another parent; planes: worst |d| = {worst_plane:.3e}; shared_trail_state (active, rebases) = {info['shared_trail']}, "
          f"mean stored window = {info['window_fraction']:.3f} of a plane")
    assert len(mism) >= 20
    if exact:
        assert sum(mism) == 0, mism
        assert worst_plane <= PLANE_EXACT, worst_plane
    else:
        assert sum(mism) <= max(2, children // 10000), mism
        assert worst_plane <= PLANE_TOL, worst_plane
    if layout != "two_shards":
        assert info["shared_trail"][0] and info["shared_trail"][1] >= 1, info     # entered by its own threshold, re-based at least once
