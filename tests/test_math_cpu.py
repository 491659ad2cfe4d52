"""The F64 likelihood's own exp / erfc / log (dbot_ros_amd/csrc/rbs_math.h) checked WITHOUT a GPU:
the header is the kernels' source and compiles for the host too (tests/cpp/math_host.cpp), so its
accuracy is pinned here against libm / mpmath, and the whole per-pixel likelihood built from it
against the oracle's (libm) pixel term, pixel by pixel.

Bars (what the float roundings that follow in the likelihood need, rbs_math.h header):
  exp   relative <= 1e-15 over [-745, 0]          erfc  absolute <= 4e-16 over [0, inf)
  log   absolute <= 2.5e-16 + 1 ulp of the result over the normal floats
  pixel term: identical to the oracle's except where a float rounding of a, b or a quotient flips
  (<= 2e-5 of pixels, each then within 2 float ulps of the sum), posterior identical except <= 2e-5
  of pixels at 1 float ulp.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
import scenarios as sc

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "cpp", "librbs_math_host.so")


@pytest.fixture(scope="module")
def mlib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "cpp"), "librbs_math_host.so"])
    return C.CDLL(LIB)


def _call(lib, fn, x, dtype=np.float64):
    x = np.ascontiguousarray(x, dtype=dtype)
    out = np.empty(x.size)
    getattr(lib, fn)(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_long(x.size))
    return out


def test_exp_nonpos(mlib):
    rng = np.random.default_rng(0)
    x = -np.concatenate([rng.uniform(0, 40, 300000), rng.uniform(0, 1, 200000), rng.uniform(0, 700, 100000),
                         [0.0, 1e-300, 0.5 * np.log(2), 708.0]])
    got, ref = _call(mlib, "rbsm_exp_nonpos", x), np.exp(x)
    assert (np.abs(got - ref) / ref).max() <= 1e-15
    # the far tail: gradual underflow and the clamp, never NaN / negative
    tail = _call(mlib, "rbsm_exp_nonpos", [-745.0, -800.0, -1e6, -np.inf])
    assert np.all(tail >= 0) and np.all(tail <= 1e-320)
    assert _call(mlib, "rbsm_exp_nonpos", [0.0])[0] == 1.0


def test_exp_against_mpmath(mlib):
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 40
    rng = np.random.default_rng(1)
    x = -rng.uniform(0, 60, 2000)
    got = _call(mlib, "rbsm_exp_nonpos", x)
    err = max(abs((mp.mpf(float(g)) - mp.exp(mp.mpf(float(v)))) / mp.exp(mp.mpf(float(v)))) for g, v in zip(got, x))
    assert err <= 6e-16


def test_erfc_pos(mlib):
    from scipy.special import erfc
    rng = np.random.default_rng(2)
    z = np.concatenate([rng.uniform(0, 6, 400000), rng.uniform(0, 0.5, 100000), rng.uniform(6, 1e3, 1000),
                        np.arange(0, 49) / 8.0, np.arange(1, 49) / 8.0 - 1e-12, [np.inf, 1e300]])
    got = _call(mlib, "rbsm_erfc_pos", z)
    assert np.abs(got - erfc(z)).max() <= 4e-16
    assert np.all(got >= 0) and np.all(got <= 1.0)
    assert 0.0 <= _call(mlib, "rbsm_erfc_pos", [np.nan])[0] <= 1.0    # NaN is clamped into the table (behaves as z = 0): never a wild read


def test_erfc_against_mpmath(mlib):
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 40
    rng = np.random.default_rng(3)
    z = rng.uniform(0, 6.5, 2000)
    got = _call(mlib, "rbsm_erfc_pos", z)
    assert max(abs(mp.mpf(float(g)) - mp.erfc(mp.mpf(float(v)))) for g, v in zip(got, z)) <= 2e-16


def test_log_f32(mlib):
    rng = np.random.default_rng(4)
    x = np.concatenate([rng.uniform(1e-3, 1e4, 400000), np.exp(rng.uniform(-87, 88, 300000)), rng.uniform(0.9, 1.1, 100000),
                        [1.0, 2.0, 0.5, np.float32(1.0) + np.finfo(np.float32).eps, 1.1754944e-38, 3.4028235e38]]).astype(np.float32)
    got, ref = _call(mlib, "rbsm_log_f32", x, np.float32), np.log(x.astype(np.float64))
    assert np.all(np.abs(got - ref) <= 2.5e-16 + np.spacing(np.abs(ref)))
    assert abs(_call(mlib, "rbsm_log_f32", [1.0], np.float32)[0]) <= 1e-17
    # not a positive normal float: what log gives (never a finite number read off the exponent field)
    odd = _call(mlib, "rbsm_log_f32", [0.0, np.inf, np.nan, -1.0, 1e-40], np.float32)
    assert odd[0] == -np.inf and odd[1] == np.inf and np.isnan(odd[2]) and np.isnan(odd[3])
    assert abs(odd[4] - np.log(np.float64(np.float32(1e-40)))) <= 1e-13


def _pixels(n, seed):
    """(observation, rendered depth, prior) triples as the raster kernel meets them: the object seen
    (|r - o| of a few sigma), occluders in front (o << r), the background behind (o >> r), priors
    over the whole unit interval."""
    rng = np.random.default_rng(seed)
    r = rng.uniform(0.3, 3.0, n)
    sigma = 0.003 + 0.0014247 * r * r
    kind = rng.integers(0, 4, n)
    o = np.where(kind <= 1, r + sigma * rng.normal(0, 1.5, n),
                 np.where(kind == 2, r - rng.uniform(0.01, 0.29, n), r + rng.uniform(0.01, 3.0, n)))
    prior = np.where(rng.random(n) < 0.5, rng.uniform(0.0, 1.0, n), np.float32(0.1))
    return o.astype(np.float32), r.astype(np.float32), prior.astype(np.float32)


@pytest.mark.parametrize("params", [{}, {"tail_weight": 0.05, "model_sigma": 0.001, "sigma_factor": 0.003}])
def test_pixel_likelihood_matches_the_oracle_pixel_by_pixel(mlib, params):
    om, cam, P = sc.make_scene(("m1_l2",), 80, 60, max_particles=1)
    for k, v in params.items():
        setattr(P.kinect, k, v)
    orc = ob.Oracle(om, cam, P, max_particles=1, mode=ob.EAGER)
    n = 1_000_000
    o, r, prior = _pixels(n, 5)
    ref_ll, ref_post = orc.pixel_terms(o, r, prior)
    ll, post = np.empty(n), np.empty(n, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    mlib.rbsm_pixel_loglik(o.ctypes.data_as(fp), r.ctypes.data_as(fp), prior.ctypes.data_as(fp), C.c_long(n),
                           C.c_double(P.kinect.tail_weight), C.c_double(P.kinect.model_sigma), C.c_double(P.kinect.sigma_factor),
                           C.c_double(np.log(2.0)), C.c_double(6.0), ll.ctypes.data_as(C.POINTER(C.c_double)), post.ctypes.data_as(fp))
    assert np.all(np.isfinite(ll)) and np.all(np.isfinite(ref_ll))
    same = ll == ref_ll
    d = np.abs(ll - ref_ll)
    # where the argument of the log (a float) is the same, the logs agree to their rounding; a flipped
    # float rounding moves the term by a float ulp or two of the ratio
    assert d[same].size and np.all(d <= 2.6e-7)
    close = d <= 2.5e-16 + 2 * np.spacing(np.abs(ref_ll))
    assert (~close).mean() <= 2e-5, (~close).mean()
    pd = np.abs(post.view(np.int32).astype(np.int64) - ref_post.view(np.int32).astype(np.int64))
    assert pd.max() <= 1 and (pd != 0).mean() <= 2e-5, (pd.max(), (pd != 0).mean())
    # a particle's sum over 5 000 such pixels: the north-star tolerance with six orders to spare
    s, sr = ll.reshape(-1, 5000).sum(1), ref_ll.reshape(-1, 5000).sum(1)
    assert (np.abs(s - sr) / np.maximum(1.0, np.abs(sr))).max() <= 1e-11
