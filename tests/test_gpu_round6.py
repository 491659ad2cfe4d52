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
