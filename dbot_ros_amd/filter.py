"""The filter step around the hot path: log-weight update,
max-shift normalisation, KL(belief || uniform) = log N - H, multinomial resampling by
upper_bound(cumsum(w), u), weighted mean.  These are the O(N) host-side scans the reference's
Rao-Blackwellised coordinate particle filter runs once per sampling block
(max_kl_divergence R:config/particle_tracker.yaml:32-36, read at
R:source/dbot_ros/tracker/particle_tracker_node.cpp:212; evaluation_count :209).

Randomness is HOST-SUPPLIED: fl's per-distribution mt19937 streams cannot be reproduced
without fl, so every function that needs uniforms takes them as an argument -- identical
uniforms on every rank give identical parents (multi-GPU, dist.py).
"""
import numpy as np


def normalized_weights(log_weights):
    lw = np.asarray(log_weights, dtype=np.float64)
    w = np.exp(lw - lw.max())
    return w / w.sum()


def kl_to_uniform(weights):
    """KL(w || uniform) = log N - H(w); 0 for uniform weights, log N for a single survivor."""
    w = np.asarray(weights, dtype=np.float64)
    nz = w[w > 0]
    return float(np.log(w.size) + (nz * np.log(nz)).sum())


def multinomial_resample(weights, uniforms):
    """Parent index of each child: upper_bound(cumsum(w), u), clipped to N-1."""
    c = np.cumsum(np.asarray(weights, dtype=np.float64))
    c /= c[-1]
    idx = np.searchsorted(c, np.asarray(uniforms, dtype=np.float64), side="right")
    return np.minimum(idx, len(c) - 1).astype(np.int32)


def weighted_mean(weights, deltas):
    """sum_i w_i * delta_i over the state vector (rotation vectors are small deltas)."""
    return np.asarray(weights, dtype=np.float64) @ np.asarray(deltas, dtype=np.float64)


class RbcFilterBlock:
    """One sampling block of the RBC particle filter for one frame:
        new_ll = sensor.loglikes(particles, indices, update)
        log_w += new_ll - ll ; ll = new_ll
        if KL(belief || uniform) > max_kl: multinomial resample
    `sensor` is anything with loglikes_poses(poses, indices, update) (product or oracle)."""

    def __init__(self, n, max_kl_divergence=2.0):
        self.n = n
        self.max_kl = max_kl_divergence
        self.log_weights = np.zeros(n)
        self.loglikes = np.zeros(n)
        self.indices = np.zeros(n, dtype=np.int32)

    def step(self, sensor, poses, uniforms, update=True):
        """Returns (parents or None, new log-likelihoods)."""
        new_ll = sensor.loglikes_poses(poses, self.indices, update=update)
        self.log_weights += new_ll - self.loglikes
        self.loglikes = new_ll
        w = normalized_weights(self.log_weights)
        if kl_to_uniform(w) > self.max_kl:
            parents = multinomial_resample(w, uniforms)
            # children inherit the parent's occlusion slot, likelihood and (uniform) weight
            self.indices = self.indices[parents].copy() if not update else parents.copy()
            self.loglikes = self.loglikes[parents]
            self.log_weights = np.zeros(self.n)
            return parents, new_ll
        return None, new_ll
