"""rbs_config.occlusion_mode = RBS_OCC_REFERENCE ("stamped planes", round 6): the device keeps the reference CPU model's own
occlusion bookkeeping -- the float posterior a pixel was last updated to and a per-pixel age -- and propagates in binary64 at
use with the oracle's operations (oracle/rbsensor_oracle.c orc_propagate; constants at
R:source/dbot_ros/tracker/particle_tracker_node.cpp:176-189).  Against the LAZY (reference-semantics) oracle the priors are
then bit-identical, and the bars below are the transcendentals' own: 1e-11 relative on log-likelihoods where the float-stepped
device rule has 1e-5, planes bit for bit."""
import numpy as np
import pytest

import oracle_binding as ob
import scenarios as sc
from dbot_ros_amd import RbSensor, RbSensorError, synth

pytestmark = pytest.mark.gpu

TOL = 1e-11   # relative to max(1, |ll|): what is left is the library's exp / erfc / log against libm's (<= 9e-16 per pixel term)


def _rel(a, b):
    return float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max())


def _planes_equal(g, lazy, slots, allow=0):
    """Device planes (effective values as of the last updating call) against the LAZY model's "as of now" planes: the same
    binary64 expression on the same stored floats, so bit for bit -- up to `allow` pixels at one float ulp (a posterior whose
    float rounding sits within 1e-16 of a boundary: rbs_math against libm)."""
    bad = 0
    for s in slots:
        a, b = g.get_occlusion(s), lazy.get_occlusion(s, now=True)
        d = a != b
        if d.any():
            assert np.abs(a[d] - b[d]).max() <= 1.2e-7, (s, float(np.abs(a[d] - b[d]).max()))
            bad += int(d.sum())
    assert bad <= allow, bad


@pytest.mark.parametrize("slab_px", [-1, 4096])
def test_sequence_is_the_lazy_oracle(slab_px):
    """A resampled sequence, 24 frames: every particle, every frame; then the planes."""
    n = 96
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=n)
    lazy = ob.Oracle(om, cam, P, max_particles=n, mode=ob.LAZY)
    frames = sc.make_frames(lazy, 1, 24, seed=3)
    ref = sc.run_sequence(lazy, frames, n)
    with RbSensor(om, cam, P, max_particles=n, occlusion="reference", slab_px=slab_px) as g:
        got = sc.run_sequence(g, frames, n)
        worst = max(_rel(a, b) for a, b in zip(got, ref))
        print(f"stamped planes vs LAZY oracle, 24 resampled frames: {worst:.2e}")
        assert worst <= TOL
        _planes_equal(g, lazy, range(0, n, 7), allow=2)


def test_skipped_frames_and_read_only_calls():
    """Frames without an updating call in between (the clock runs on), read-only calls at any point: ages, not steps."""
    n = 48
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=n)
    lazy = ob.Oracle(om, cam, P, max_particles=n, mode=ob.LAZY)
    frames = sc.make_frames(lazy, 1, 10, seed=5)
    with RbSensor(om, cam, P, max_particles=n, occlusion="reference") as g:
        rng = np.random.default_rng(2)
        for s in (lazy, g):
            s.reset()
        idx_o, idx_g = np.zeros(n, np.int32), np.zeros(n, np.int32)
        for k, (truth, frame) in enumerate(frames):
            for s in (lazy, g):
                s.set_observation(frame)
            poses = synth.particle_poses(truth, n, rng, scale=1.0 + 0.3 * k)
            update = k % 3 != 1          # every third frame is only looked at
            a = g.loglikes_poses(poses, idx_g, update=update)
            b = lazy.loglikes_poses(poses, idx_o, update=update)
            assert _rel(a, b) <= TOL, (k, _rel(a, b))
            if update:
                w = np.exp(b - b.max())
                idx_o = np.sort(rng.choice(n, size=n, p=w / w.sum())).astype(np.int32)
                idx_g = idx_o.copy()
            if k == 6:   # two more frames pass unseen
                for s in (lazy, g):
                    s.set_observation(frame)
                    s.set_observation(frame)


def test_mode_needs_f64_and_windows():
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=8)
    with pytest.raises(RbSensorError):
        RbSensor(om, cam, P, max_particles=8, occlusion="reference", precision="f32")
    with pytest.raises(RbSensorError):
        RbSensor(om, cam, P, max_particles=8, occlusion="reference", state_layout="dense")
    # ... and a process that forgets within the 16-bit age counter's range (the reference's constants: within 1 628 frames)
    import copy
    slow = copy.deepcopy(P)
    slow.occlusion.p_occluded_occluded, slow.occlusion.p_occluded_visible = 0.9995, 0.0005
    with pytest.raises(RbSensorError, match="16-bit age"):
        RbSensor(om, cam, slow, max_particles=8, occlusion="reference")
    with RbSensor(om, cam, slow, max_particles=8, occlusion="device"):
        pass


@pytest.mark.parametrize("slab_px", [0, 16384])
def test_plane_transport_against_a_shared_plane(monkeypatch, slab_px):
    """Stamped planes stored against the shared background plane: a plane sent with rbs_export_window and taken in by
    rbs_import_window is the same plane -- also where it merely EQUALS the shared plane's values inside the window (found by running
    the whole suite with RBS_OCC=reference: such pixels had been marked "background", which inside a window means the scalar level)."""
    import torch
    monkeypatch.setenv("RBS_SHARED_TRAIL", "1")
    monkeypatch.setenv("RBS_STP_ENTER", "0.0")
    monkeypatch.setenv("RBS_STP_EVERY", "3")
    n, cols, rows = 48, 320, 240
    om, cam, P = sc.make_scene(("m1_l2",), cols, rows, max_particles=n)
    lazy = ob.Oracle(om, cam, P, max_particles=1, mode=ob.LAZY)
    rng = np.random.default_rng(31)
    frames = []
    for k in range(14):
        t = synth.truth_pose(1, frame=k)
        t[:, 9] += -0.08 + 0.012 * k
        frames.append((t, synth.make_frame(lazy.render_depth(t), rows, cols, rng)))
    poses = [synth.particle_poses(t, n, rng, scale=1.0) for t, _ in frames]
    parents = [np.sort(rng.choice(n, size=n, p=(lambda w: w / w.sum())(rng.random(n) ** 8))).astype(np.int32) for _ in frames]
    with RbSensor(om, cam, P, max_particles=n, occlusion="reference", slab_px=slab_px) as g:
        g.set_timing_every(1)
        g.reset()
        idx = np.zeros(n, np.int32)
        for k, (_, frame) in enumerate(frames):
            g.set_observation(frame)
            g.loglikes_poses(poses[k], idx, update=True)
            idx = parents[k].copy()
        assert g.shared_trail_state()[0]
        buf = torch.empty(cols * rows, dtype=torch.float32, device="cuda:0")
        for src in (3, 17):
            before = g.get_occlusion(src)
            rect = g.export_window(src, buf.data_ptr(), buf.numel())
            assert rect == (0, 0, cols, rows)                      # against a shared plane the plane travels whole
            assert np.array_equal(buf.cpu().numpy(), before)
            g.import_window(n - 1, rect, buf.data_ptr())
            g.synchronize()
            assert np.array_equal(g.get_occlusion(n - 1), before), src
            g.set_occlusion(n - 2, before)
            assert np.array_equal(g.get_occlusion(n - 2), before), src


def test_the_parity_suite_in_reference_mode():
    """tests/test_gpu_parity.py's windowed cases once more with the LIBRARY in occlusion_mode REFERENCE and the tests' device-rule
    oracles replaced by the reference-semantics one (RBS_OCC=reference: the library takes it where the caller leaves the mode
    open, tests/oracle_binding.py switches the oracle): sequences, several bodies, read-only calls, skipped frames, NaN / inf frames,
    off-screen particles, bad parent slots, many tiles, varying particle counts and call patterns, threads, borrowed frames -- at
    those tests' own bars (1e-9 on log-likelihoods, planes bit for bit up to 1e-4 of pixels at one ulp).  Left out: the cases that test
    the device rule itself (goldens generated by it, its background snap, the wide-window route) or compare with dense planes."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, RBS_OCC="reference")
    sel = "window and not golden and not windows_follow and not wide_windows and not layouts_hold and not 322 and not eager_background"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_parity.py"), "-q", "-x", "-m", "gpu", "-k", sel,
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=1200, env=env, cwd=os.path.dirname(here))
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    print("\ntest_gpu_parity.py (windowed cases) with RBS_OCC=reference: " + tail)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]
    assert " passed" in tail and "failed" not in tail


@pytest.mark.parametrize("slab_px", [-1, 0, 2048])      # (2 048: slabs that must grow several times on the way -- values AND ages move)
def test_large_windows_without_the_shared_trail(slab_px):
    """An object that crosses the image with the shared trail switched off (rbs_set_option): the windows grow to most of the plane -- the
    regime where the device rule changes to its whole-plane machinery, which stamped planes do not have (their windowed kernels carry
    on) -- and shrink again behind the object only as ages pass age_max.  Every particle, every frame against the LAZY oracle."""
    n, cols, rows = 64, 160, 120
    om, cam, P = sc.make_scene(("m1_l2",), cols, rows, max_particles=n)
    lazy = ob.Oracle(om, cam, P, max_particles=n, mode=ob.LAZY)
    rng = np.random.default_rng(41)
    with RbSensor(om, cam, P, max_particles=n, occlusion="reference", slab_px=slab_px) as g:
        g.set_option("shared_trail", 0)
        g.set_option("timing_every", 1)
        g.reset(); lazy.reset()
        idx_g, idx_o = np.zeros(n, np.int32), np.zeros(n, np.int32)
        frac = 0.0
        for k in range(48):
            t = synth.truth_pose(1, frame=k)
            t[:, 9] += -0.27 + 0.0115 * k           # corner to corner across the image
            t[:, 10] += -0.19 + 0.0082 * k
            frame = synth.make_frame(lazy.render_depth(t), rows, cols, rng)
            g.set_observation(frame); lazy.set_observation(frame)
            poses = synth.particle_poses(t, n, rng, scale=1.0)
            a, b = g.loglikes_poses(poses, idx_g, update=True), lazy.loglikes_poses(poses, idx_o, update=True)
            assert _rel(a, b) <= TOL, (k, _rel(a, b))
            p = np.sort(rng.choice(n, size=n, p=(lambda w: w / w.sum())(rng.random(n) ** 6))).astype(np.int32)
            idx_g, idx_o = p.copy(), p.copy()
            frac = max(frac, g.window_fraction())
        assert not g.shared_trail_state()[0]
        area = np.mean([max(0, w[2] - w[0]) * max(0, w[3] - w[1]) for w in (g.get_window(q) for q in range(0, n, 5))]) / (rows * cols)
        print(f"\nslab_px {slab_px}: stored window at the end {area:.2f} of the plane (sampled fraction {frac:.2f})")
        assert area > 0.5
        _planes_equal(g, lazy, range(0, n, 9), allow=2)


def test_the_layout_suites_in_reference_mode():
    """... and the suites of the storage layouts and of the handles that read each other's planes -- slabs (tests/test_gpu_slabs.py), several
    devices in one handle (test_gpu_multidevice.py), attached processes (test_gpu_peers.py), the shared trail and the host routes of round 5
    (test_gpu_round5.py) -- once more with the library in occlusion_mode REFERENCE and the reference-semantics oracle as the checker
    (RBS_OCC=reference, as above).  Left out: whole-plane (dense) cases, which the mode does not have."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, RBS_OCC="reference")
    files = [os.path.join(here, f) for f in ("test_gpu_slabs.py", "test_gpu_multidevice.py", "test_gpu_peers.py", "test_gpu_round5.py")]
    r = subprocess.run([sys.executable, "-m", "pytest"] + files + ["-q", "-x", "-m", "gpu", "-k", "not dense", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=2400, env=env, cwd=os.path.dirname(here))
    tail = [l for l in r.stdout.strip().splitlines() if " passed" in l or " failed" in l]
    print("\nslabs / multi-device / peers / round-5 suites with RBS_OCC=reference: " + (tail[-1] if tail else ""))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]
    assert tail and "failed" not in tail[-1]
