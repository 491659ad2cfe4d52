"""Filter step: weights, KL, multinomial resampling, and the block loop run
against the oracle."""
import numpy as np

import oracle_binding as ob
import scenarios as sc
from dbot_ros_amd import filter as flt
from dbot_ros_amd import synth


def test_weights_and_kl():
    w = flt.normalized_weights(np.array([1000.0, 1000.0, 1000.0, 1000.0]))
    assert np.allclose(w, 0.25) and abs(flt.kl_to_uniform(w)) < 1e-15
    w = flt.normalized_weights(np.array([0.0, -1e9, -1e9]))
    assert np.allclose(w, [1, 0, 0]) and np.isclose(flt.kl_to_uniform(w), np.log(3))
    lw = np.random.default_rng(0).normal(size=100) * 3
    w = flt.normalized_weights(lw)
    assert np.isclose(w.sum(), 1.0) and np.allclose(w, np.exp(lw) / np.exp(lw).sum())
    assert np.isclose(flt.kl_to_uniform(w), (w * np.log(w * 100)).sum())


def test_multinomial_resample_is_upper_bound():
    w = np.array([0.1, 0.0, 0.4, 0.5])
    u = np.array([0.0, 0.0999, 0.1, 0.4999, 0.5, 0.99999, 1.0])
    # python restatement of std::upper_bound over the cumulative sum
    c = np.cumsum(w)
    ref = [min(next((i for i, x in enumerate(c) if x > v), len(c) - 1), len(c) - 1) for v in u]
    assert flt.multinomial_resample(w, u).tolist() == ref
    assert 1 not in flt.multinomial_resample(w, np.random.default_rng(1).random(1000))  # zero weight never drawn
    big = flt.multinomial_resample(w, np.random.default_rng(2).random(200000))
    assert np.allclose(np.bincount(big, minlength=4) / 200000, w, atol=5e-3)


def test_filter_block_on_oracle_tracks_truth():
    """Three frames of propagate-free filtering: the block resamples when the weights
    concentrate, parents are a valid multiset, weights reset, indices follow the parents."""
    n = 48
    om, cam, P = sc.make_scene(("m1_l2",), 80, 60, max_particles=n)
    o = ob.Oracle(om, cam, P, max_particles=n, mode=ob.LAZY)
    frames = sc.make_frames(o, 1, 3, seed=2)
    rng = np.random.default_rng(4)
    blk = flt.RbcFilterBlock(n, max_kl_divergence=2.0)
    o.reset()
    resampled = 0
    for truth, frame in frames:
        o.set_observation(frame)
        poses = synth.particle_poses(truth, n, rng, scale=2.0)
        parents, ll = blk.step(o, poses, rng.random(n), update=True)
        assert np.isfinite(ll).all()
        if parents is not None:
            resampled += 1
            assert parents.min() >= 0 and parents.max() < n
            assert (blk.indices == parents).all() and np.allclose(blk.log_weights, 0)
            # the best particle survives with overwhelming probability
            assert int(np.argmax(ll)) in parents
        else:
            assert (blk.indices == np.arange(n)).all()
    assert resampled >= 1
