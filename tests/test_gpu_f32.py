"""Likelihood precision F32 (rbs_config.likelihood_precision = RBS_PRECISION_F32) against the CPU
oracle.  Coverage, depth and the occlusion process are precision independent and stay bit-exact;
the per-pixel likelihood runs in float32, so the bars here are

  * log-likelihood vs the LAZY oracle (reference CPU semantics)  |d| <= 1e-5 * max(1,|ll|)
    -- BASELINE.json north_star's tolerance -- for every particle whose sum is well conditioned,
    |ll| >= 0.1 * S with S = the sum of the MAGNITUDES of the per-pixel terms it adds up (the
    oracle reports S); and for EVERY particle |d| <= 1e-6 * max(1, S): a log-likelihood is a sum
    of ~5 000 terms of magnitude ~3, float32 arithmetic bounds its error relative to S
    (measured 4e-8 * S, i.e. ~5e-4 absolute on sums of magnitude 1e4), not relative to a sum that
    happens to cancel to nearly zero;
  * log-likelihood vs the EAGER oracle and vs the product's own F64 mode: the same bars;
  * occlusion planes vs the EAGER oracle: |d| <= 2e-6 absolute (posteriors are float32 quotients
    of float32 terms; untouched and never-covered pixels stay bit-exact).
"""
import numpy as np
import pytest

import oracle_binding as ob
import scenarios as sc
from dbot_ros_amd import RbSensor, synth

pytestmark = pytest.mark.gpu

TOL = 1e-5
PLANE_TOL = 2e-6


@pytest.fixture(autouse=True, params=["window", "dense"])
def state_layout(request):
    return request.param


def rel_err(a, b):
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


def assert_f32_close(ll, ref, S, what=""):
    d = np.abs(ll - ref)
    well = np.abs(ref) >= 0.1 * S
    assert (d[well] <= TOL * np.maximum(1.0, np.abs(ref[well]))).all(), (what, rel_err(ll, ref)[well].max())
    assert (d <= 1e-6 * np.maximum(1.0, S)).all(), (what, (d / np.maximum(1.0, S)).max())
    return (rel_err(ll, ref)[well].max() if well.any() else 0.0), (d / np.maximum(1.0, S)).max()


@pytest.mark.parametrize("meshes,cols,rows,n", [(("m1",), 640, 480, 24), (("m3",), 640, 480, 16),
                                                 (("m1",), 80, 60, 64), (("box12",), 640, 480, 8),
                                                 (("m1", "m2", "m3"), 640, 480, 12),
                                                 (("m1_l2",), 322, 241, 16),
                                                 (("m4",), 1280, 960, 4), (("m1", "m4", "m2"), 640, 480, 6)])
def test_f32_sequence_matches_oracle(gpu_lib, state_layout, meshes, cols, rows, n):
    om, cam, P = sc.make_scene(meshes, cols, rows, max_particles=n)
    nb = len(meshes)
    eager = ob.Oracle(om, cam, P, max_particles=n, mode=ob.EAGER)
    lazy = ob.Oracle(om, cam, P, max_particles=n, mode=ob.LAZY)
    frames = sc.make_frames(eager, nb, 4, seed=3)
    with RbSensor(om, cam, P, max_particles=n, precision="f32", state_layout=state_layout) as g, \
            RbSensor(om, cam, P, max_particles=n, precision="f64", state_layout=state_layout) as g64:
        ll_g = sc.run_sequence(g, frames, n, n_bodies=nb)
        ll_64 = sc.run_sequence(g64, frames, n, n_bodies=nb)
        ll_e = sc.run_sequence(eager, frames, n, n_bodies=nb)
        S = []
        ll_l = sc.run_sequence(lazy, frames, n, n_bodies=nb, abs_sums=S)
        worst, worst_s = 0.0, 0.0
        for k in range(len(frames)):
            assert np.isfinite(ll_g[k]).all()
            for ref in (ll_l[k], ll_e[k], ll_64[k]):
                a, b = assert_f32_close(ll_g[k], ref, S[k], (meshes, k))
                worst, worst_s = max(worst, a), max(worst_s, b)
        print(f"f32 {meshes} {cols}x{rows}: worst error {worst:.2e} relative to |ll| (well-conditioned particles), "
              f"{worst_s:.2e} relative to the sum of term magnitudes")
        for slot in range(n):
            pg, pe = g.get_occlusion(slot), eager.get_occlusion(slot)
            assert np.abs(pg - pe).max() <= PLANE_TOL, np.abs(pg - pe).max()
            assert np.abs(pg - lazy.get_occlusion(slot, now=True)).max() <= 4e-6


def test_f32_long_sequence_vs_reference_semantics(gpu_lib, state_layout):
    """120 frames at 80x60 (the reference's default resolution), resampling every frame."""
    n = 32
    om, cam, P = sc.make_scene(("m1_l2",), 80, 60, max_particles=n)
    lazy = ob.Oracle(om, cam, P, max_particles=n, mode=ob.LAZY)
    frames = sc.make_frames(lazy, 1, 120, seed=11)
    with RbSensor(om, cam, P, max_particles=n, precision="f32", state_layout=state_layout) as g:
        ll_g = sc.run_sequence(g, frames, n)
    S = []
    ll_l = sc.run_sequence(lazy, frames, n, abs_sums=S)
    worst = [assert_f32_close(a, b, s_, k) for k, (a, b, s_) in enumerate(zip(ll_g, ll_l, S))]
    print(f"f32, 120 frames vs the LAZY oracle: worst {max(w[0] for w in worst):.2e} relative to |ll|, "
          f"{max(w[1] for w in worst):.2e} relative to the sum of term magnitudes")


def test_f32_depth_is_precision_independent(gpu_lib, state_layout):
    om, cam, P = sc.make_scene(("m3",), 640, 480, max_particles=2)
    rng = np.random.default_rng(5)
    with RbSensor(om, cam, P, max_particles=2, precision="f32") as a, RbSensor(om, cam, P, max_particles=2, precision="f64") as b:
        for k in range(3):
            pose = synth.particle_poses(synth.truth_pose(1, frame=3 * k), 1, rng, scale=4.0)[0]
            assert np.array_equal(a.render_depth(pose).view(np.uint32), b.render_depth(pose).view(np.uint32))
