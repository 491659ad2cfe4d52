"""Round 6: the two launch forms of the likelihood call mixed on one handle.

A borrowed frame (rbs_set_observation_borrowed) makes the next likelihood call launch as TWO kernels (geometry, then likelihood:
the frame travels in between); any other call launches as one.  The likelihood kernel draws its work items from a ticket counter of
its own, and a handle mixes the two forms call by call -- with more work items than that kernel has blocks (4 per CU) a counter left
non-zero by an earlier two-kernel call of the same parity would skip items.  The sequences below are the ones that found it."""
import numpy as np
import pytest

import oracle_binding as ob
import scenarios as sc
from dbot_ros_amd import RbSensor, synth

pytestmark = pytest.mark.gpu


def _frames(om, cam, P, nb, count, rows, cols, seed):
    o = ob.Oracle(om, cam, P, max_particles=1, mode=ob.EAGER)
    rng = np.random.default_rng(seed)
    out = []
    for k in range(count):
        t = synth.truth_pose(nb, frame=k)
        out.append((t, synth.make_frame(o.render_depth(t), rows, cols, rng)))
    o.close()
    return out


@pytest.mark.parametrize("occlusion", ["device", "reference"])
def test_borrowed_and_plain_calls_alternate_above_the_eval_grid(occlusion):
    """2 400 particles (more items than the likelihood kernel's 4 x CU blocks): borrowed, plain, borrowed, borrowed, plain, plain,
    borrowed ... against a handle that never borrows -- every log-likelihood and the planes bit for bit."""
    n, cols, rows = 2400, 320, 240
    om, cam, P = sc.make_scene(("m1_l2",), cols, rows, max_particles=n)
    frames = _frames(om, cam, P, 1, 12, rows, cols, seed=3)
    rng = np.random.default_rng(4)
    poses = [synth.particle_poses(t, n, rng, scale=1.0) for t, _ in frames]
    parents = [np.sort(rng.choice(n, size=n)).astype(np.int32) for _ in frames]
    pattern = [1, 0, 1, 1, 0, 0, 1, 0, 1, 1, 1, 0]      # 1: the frame is borrowed (two-kernel launch)
    with RbSensor(om, cam, P, max_particles=n, occlusion=occlusion) as plain, RbSensor(om, cam, P, max_particles=n, occlusion=occlusion) as g:
        plain.reset(); g.reset()
        ip, ig = np.zeros(n, np.int32), np.zeros(n, np.int32)
        for k, (_, frame) in enumerate(frames):
            f64 = np.ascontiguousarray(frame, dtype=np.float64)
            plain.set_observation(f64)
            if pattern[k]:
                g.set_observation_borrowed(f64)
            else:
                g.set_observation(f64)
            a, b = g.loglikes_poses(poses[k], ig, update=True), plain.loglikes_poses(poses[k], ip, update=True)
            assert np.isfinite(a).all()
            assert np.array_equal(a, b), (k, int((a != b).sum()), float(np.abs(a - b).max()))
            ig, ip = parents[k].copy(), parents[k].copy()
        for q in range(0, n, 211):
            assert np.array_equal(g.get_occlusion(q), plain.get_occlusion(q)), q


def test_two_bodies_borrowed_sequence():
    """Two bodies = two sampling blocks per frame (a read-only call, then the updating one): an EVEN number of calls per frame, the
    first of them two-kernel (it stages the borrowed frame), the second one-kernel -- the parity pattern of the defect."""
    n, cols, rows = 1500, 320, 240
    om, cam, P = sc.make_scene(("m1_l2", "box12"), cols, rows, max_particles=n)
    frames = _frames(om, cam, P, 2, 10, rows, cols, seed=8)
    rng = np.random.default_rng(9)
    poses = [synth.particle_poses(t, n, rng, scale=1.0) for t, _ in frames]
    parents = [np.sort(rng.choice(n, size=n)).astype(np.int32) for _ in frames]
    with RbSensor(om, cam, P, max_particles=n) as plain, RbSensor(om, cam, P, max_particles=n) as g:
        plain.reset(); g.reset()
        ip, ig = np.zeros(n, np.int32), np.zeros(n, np.int32)
        for k, (_, frame) in enumerate(frames):
            f64 = np.ascontiguousarray(frame, dtype=np.float64)
            plain.set_observation(f64)
            g.set_observation_borrowed(f64)
            for upd in (False, True):
                a, b = g.loglikes_poses(poses[k], ig if upd else ig.copy(), update=upd), plain.loglikes_poses(poses[k], ip if upd else ip.copy(), update=upd)
                assert np.array_equal(a, b), (k, upd, int((a != b).sum()))
            ig, ip = parents[k].copy(), parents[k].copy()
        for q in range(0, n, 173):
            assert np.array_equal(g.get_occlusion(q), plain.get_occlusion(q)), q


# ---- the shared trail where others read a handle's planes in place ----
TOL_EAGER = 1e-9


@pytest.mark.parametrize("occlusion,slab,precision", [("device", 0, "f64"), ("device", 16384, "f64"), ("reference", 0, "f64"), ("device", 0, "f32")])
def test_shared_trail_on_two_shards_stores_the_same_planes(monkeypatch, occlusion, slab, precision):
    """One handle over two shards (device_ids = [0, 0]): the group takes ONE decision per call for both shards, each keeps its own
    identical copy of the shared plane, re-based in the same call on the same global slot (read from its owner).  Against the same
    handle with the shared trail disabled: log-likelihoods and planes bit for bit, read-only calls included; windows smaller."""
    n, cols, rows = 128, 640, 480
    om, cam, P = sc.make_scene(("m1",), cols, rows, max_particles=n)
    o = ob.Oracle(om, cam, P, max_particles=n, mode=ob.EAGER if occlusion == "device" else ob.LAZY)
    rng = np.random.default_rng(31)
    frames = []
    for k in range(26):
        t = synth.truth_pose(1, frame=k)
        t[:, 9] += -0.12 + 0.01 * k
        t[:, 10] += -0.06 + 0.005 * k
        frames.append((t, synth.make_frame(o.render_depth(t), rows, cols, rng)))
    poses = [synth.particle_poses(t, n, rng, scale=1.0) for t, _ in frames]
    parents = [np.sort(rng.choice(n, size=n, p=(lambda w: w / w.sum())(rng.random(n) ** 8))).astype(np.int32) for _ in frames]
    monkeypatch.setenv("RBS_SHARED_TRAIL", "0")
    with RbSensor(om, cam, P, max_particles=n, device_ids=[0, 0], occlusion=occlusion, slab_px=slab, precision=precision) as plain:
        monkeypatch.setenv("RBS_SHARED_TRAIL", "1")
        monkeypatch.setenv("RBS_STP_ENTER", "0.0")
        monkeypatch.setenv("RBS_STP_EVERY", "3")
        with RbSensor(om, cam, P, max_particles=n, device_ids=[0, 0], occlusion=occlusion, slab_px=slab, precision=precision) as g:
            g.set_timing_every(1); plain.set_timing_every(1)
            for s_ in (g, plain, o):
                s_.reset()
            ig, ip, io = (np.zeros(n, np.int32) for _ in range(3))
            for k, (_, frame) in enumerate(frames):
                for s_ in (g, plain, o):
                    s_.set_observation(frame)
                if k % 5 == 4:
                    ra, rb = g.loglikes_poses(poses[k - 1], ig.copy(), update=False), plain.loglikes_poses(poses[k - 1], ip.copy(), update=False)
                    assert np.array_equal(ra, rb), (k, np.abs(ra - rb).max())
                la, lb = g.loglikes_poses(poses[k], ig, update=True), plain.loglikes_poses(poses[k], ip, update=True)
                lo = o.loglikes_poses(poses[k], io, update=True)
                assert np.array_equal(la, lb), (k, np.abs(la - lb).max())
                assert (np.abs(la - lo) / np.maximum(1.0, np.abs(lo))).max() <= (TOL_EAGER if precision == "f64" else 1e-3)
                ig, ip, io = parents[k].copy(), parents[k].copy(), parents[k].copy()
            active, rebases = g.shared_trail_state()
            assert active and rebases >= 3, (active, rebases)
            assert plain.shared_trail_state() == (False, 0)
            area = lambda w: max(0, w[2] - w[0]) * max(0, w[3] - w[1])
            slots = list(range(0, n, max(1, n // 16)))
            a_g, a_p = np.mean([area(g.get_window(q)) for q in slots]), np.mean([area(plain.get_window(q)) for q in slots])
            print(f"\ntwo shards, occlusion {occlusion}, slab {slab}: mean window: shared trail {a_g:.0f} px, scalar background {a_p:.0f} px ({rebases} re-basings)")
            assert a_g < 0.75 * a_p
            for q in slots:
                assert np.array_equal(g.get_occlusion(q), plain.get_occlusion(q)), q


def test_shared_trail_across_processes():
    """tests/peer_trail_worker.py: two processes on cuda:0, handles attached over HIP IPC, dist.PeerShardedStep with
    shared_trail=True (rbs_shared_trail_rebase on every rank before the same step) against the same job without it."""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "peer_trail_worker.py")
    port = 27100 + os.getpid() % 800
    r = subprocess.run([sys.executable, script, str(port)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "TRAIL_OK" in r.stdout, r.stdout[-2000:]
    print(r.stdout[-900:])


def test_set_option_switches_the_shared_trail_and_refuses_nonsense():
    """rbs_set_option: the switches a caller may need as an API (the environment variables of the same meaning are tooling).  The
    shared trail switched off by option never activates on a sequence that activates it otherwise; values are the same bits."""
    from dbot_ros_amd import RbSensorError
    n, cols, rows = 64, 320, 240
    om, cam, P = sc.make_scene(("m1_l2",), cols, rows, max_particles=n)
    frames = []
    o = ob.Oracle(om, cam, P, max_particles=1, mode=ob.EAGER)
    rng = np.random.default_rng(2)
    for k in range(20):
        t = synth.truth_pose(1, frame=k)
        t[:, 9] += -0.10 + 0.012 * k
        frames.append((t, synth.make_frame(o.render_depth(t), rows, cols, rng)))
    o.close()
    poses = [synth.particle_poses(t, n, rng, scale=1.0) for t, _ in frames]
    parents = [np.sort(rng.choice(n, size=n, p=(lambda w: w / w.sum())(rng.random(n) ** 8))).astype(np.int32) for _ in frames]
    with RbSensor(om, cam, P, max_particles=n) as a, RbSensor(om, cam, P, max_particles=n) as b:
        for s_ in (a, b):
            s_.set_option("timing_every", 1)
            s_.set_option("shared_trail_enter", 0.02)
            s_.set_option("shared_trail_every", 3)
        b.set_option("shared_trail", 0)
        a.reset(); b.reset()
        ia, ib = np.zeros(n, np.int32), np.zeros(n, np.int32)
        for k, (_, frame) in enumerate(frames):
            a.set_observation(frame); b.set_observation(frame)
            la, lb = a.loglikes_poses(poses[k], ia, update=True), b.loglikes_poses(poses[k], ib, update=True)
            assert np.array_equal(la, lb), k
            ia, ib = parents[k].copy(), parents[k].copy()
        assert a.shared_trail_state()[0] and a.shared_trail_state()[1] >= 2
        assert b.shared_trail_state() == (False, 0)
        for q in range(0, n, 7):
            assert np.array_equal(a.get_occlusion(q), b.get_occlusion(q))
        for name, bad in (("shared_trail_enter", 0.0), ("shared_trail_every", 0), ("tracker_split_max", -1), ("timing_every", 0)):
            with pytest.raises(RbSensorError):
                a.set_option(name, bad)
        with pytest.raises(RbSensorError):
            a._check(a._lib.rbs_set_option(a._h, 99, 1.0))


def test_pixel_centre_convention_through_a_shifted_principal_point():
    """DESIGN.md section 2 / INTEGRATION.md section 3: were upstream's renderer to sample at pixel centres, the binding hands K over with
    (cx - 0.5, cy - 0.5).  The DEVICE with the shifted K against the oracle VARIANT that samples at centres under the unshifted K:
    rendered depth bit for bit, a resampled sequence's log-likelihoods at the usual bar."""
    import copy
    n, cols, rows = 64, 320, 240
    om, cam, P = sc.make_scene(("m1_l2",), cols, rows, max_particles=n)
    cam_shift = copy.deepcopy(cam)
    cam_shift.camera_matrix = np.array(cam.camera_matrix, dtype=np.float64).copy()
    cam_shift.camera_matrix[0, 2] -= 0.5
    cam_shift.camera_matrix[1, 2] -= 0.5
    centres = ob.Oracle(om, cam, P, max_particles=n, mode=ob.LAZY, variant="cov_centres")
    rng = np.random.default_rng(12)
    with RbSensor(om, cam_shift, P, max_particles=n, occlusion="reference") as g:
        g.reset(); centres.reset()
        idx_g, idx_o = np.zeros(n, np.int32), np.zeros(n, np.int32)
        for k in range(8):
            truth = synth.truth_pose(1, frame=k)
            d_o, d_g = centres.render_depth(truth), g.render_depth(truth)
            assert np.array_equal(d_o, d_g, equal_nan=True), k
            frame = synth.make_frame(d_o, rows, cols, rng)
            g.set_observation(frame); centres.set_observation(frame)
            poses = synth.particle_poses(truth, n, rng, scale=1.0)
            a, b = g.loglikes_poses(poses, idx_g, update=True), centres.loglikes_poses(poses, idx_o, update=True)
            assert (np.abs(a - b) / np.maximum(1.0, np.abs(b))).max() <= 1e-10, k
            p = np.sort(rng.choice(n, size=n)).astype(np.int32)
            idx_g, idx_o = p.copy(), p.copy()
    centres.close()
