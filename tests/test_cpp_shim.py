"""The C++ flavour of the plugin-surface mirror (include/dbot_amd/rb_sensor_builder.hpp),
driven the way the reference's node builds and uses the sensor."""
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
import scenarios as sc
from dbot_ros_amd import CameraData, ObjectModel, RbSensorBuilder, pose, synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BIN = os.path.join(HERE, "cpp", "shim_check")


BIN_ASAN = os.path.join(HERE, "cpp", "shim_check_asan")
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:protect_shadow_gap=0:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"}


def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "cpp")])


def _build_asan():
    """The sanitizer build is best effort (__graft_entry__.build()): a toolchain without the
    sanitizer runtimes skips these cases instead of failing them."""
    try:
        subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "..", "dbot_ros_amd", "csrc"), "asan"])
        subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "cpp"), "asan"])
    except (subprocess.CalledProcessError, OSError) as e:
        pytest.skip(f"no ASan/UBSan build of the library with this toolchain: {e}")


def _clean(stderr):
    bad = [ln for ln in stderr.splitlines() if "AddressSanitizer" in ln or "runtime error" in ln or "SUMMARY" in ln]
    assert not bad, "\n".join(bad[:20])


def _scene(tmp_path):
    rows, cols, n = 60, 80, 6
    v1, f1 = synth.mesh_m1(level=2)
    v2, f2 = synth.mesh_box12()
    K = synth.camera_matrix(cols, rows)
    rng = np.random.default_rng(3)
    default = np.zeros(24)
    default[0:3], default[3:6] = (-0.05, 0.0, 0.6), (0.2, -0.4, 0.1)
    default[12:15], default[15:18] = (0.06, 0.01, 0.65), (0.5, 0.1, -0.3)
    deltas = np.zeros((n, 24))
    for b in range(2):
        deltas[:, 12 * b:12 * b + 3] = rng.normal(0, 0.004, (n, 3))
        deltas[:, 12 * b + 3:12 * b + 6] = rng.normal(0, 0.03, (n, 3))
    # frame from the oracle's renderer at the default pose
    om = ObjectModel([v1, v2], [f1, f2], center=True)
    cam = CameraData(K, rows, cols)
    P = RbSensorBuilder.Parameters(sample_count=n)
    o = ob.Oracle(om, cam, P, max_particles=n, mode=ob.EAGER)
    frame = synth.make_frame(o.render_depth(pose.compose_with_default(np.zeros((1, 24)), default, 2)[0]),
                             rows, cols, rng).astype(np.float64)
    path = tmp_path / "scene.txt"
    with open(path, "w") as f:
        f.write(f"{rows} {cols}\n" + " ".join(repr(float(x)) for x in K.ravel()) + "\n2\n")
        for v, t in ((v1, f1), (v2, f2)):
            f.write(f"{len(v)} {len(t)}\n" + " ".join(repr(float(x)) for x in v.ravel()) + "\n")
            f.write(" ".join(str(int(x)) for x in t.ravel()) + "\n")
        f.write(f"{n}\n" + " ".join(repr(float(x)) for x in default) + "\n")
        for d in deltas:
            f.write(" ".join(repr(float(x)) for x in d) + "\n")
        f.write(" ".join("nan" if np.isnan(x) else repr(float(x)) for x in frame) + "\n")
    return path, o, default, deltas, frame, n


def test_dbot_binding_compiles():
    """integration/dbot/rb_sensor_mi355x.h -- the RbSensor<State> subclass a dbot maintainer adds -- type-checks against
    STAND-IN dbot / fl / Eigen headers (tests/cpp/stubs/, labelled as such; the real ones are not in the build image), with
    the include names the reference itself uses (<dbot/...>.h, R:source/dbot_ros/tracker/particle_tracker_node.cpp:22-26):
    the file cannot rot unnoticed.  Compile-only."""
    import subprocess
    root = ROOT
    binding = os.path.join(root, "integration", "dbot", "rb_sensor_mi355x.h")
    txt = open(binding).read()
    assert "#include <dbot/camera_data.h>" in txt and "#include <dbot/object_model.h>" in txt and "#include <dbot/model/rb_sensor.h>" in txt
    assert ".hpp>" not in txt
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wextra", "-I" + os.path.join(root, "tests", "cpp", "stubs"),
                        "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "integration"),
                        os.path.join(root, "tests", "cpp", "dbot_binding_check.cpp")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    # INTEGRATION.md section 2 shows this very file (not a paraphrase that can drift)
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    body = txt[txt.index("#pragma once"):]
    assert body.strip() in md, "INTEGRATION.md section 2 must quote integration/dbot/rb_sensor_mi355x.h verbatim (tools/sync_integration_md.py)"


def test_cpp_shim_builds_and_fails_loudly_without_gpu(tmp_path):
    _build()
    from dbot_ros_amd import _capi
    if _capi.load().rbs_device_count() > 0:
        pytest.skip("GPU visible: covered by the gpu test")
    path, *_ = _scene(tmp_path)
    out = subprocess.run([BIN, str(path)], capture_output=True, text=True, check=True).stdout
    assert out.startswith("NO_DEVICE") and "no CPU path" in out


def test_host_code_under_sanitizers_without_gpu(tmp_path):
    """the library's host code (argument checks, error paths, the C++ mirror's
    exception translation) under AddressSanitizer + UndefinedBehaviorSanitizer, on the path a box
    without a GPU takes."""
    _build_asan()
    from dbot_ros_amd import _capi
    if _capi.load().rbs_device_count() > 0:
        pytest.skip("GPU visible: covered by the gpu test")
    path, *_ = _scene(tmp_path)
    r = subprocess.run([BIN_ASAN, str(path)], capture_output=True, text=True, env=dict(os.environ, **SAN_ENV))
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.startswith("NO_DEVICE")
    _clean(r.stderr)


@pytest.mark.gpu
def test_host_code_under_sanitizers(tmp_path, gpu_lib):
    """The same driver as test_cpp_shim_matches_oracle -- sensor construction in the node's order,
    two loglikes calls, three frames of the device tracker -- against the ASan + UBSan build of the
    library's host code: no report, and the very same output as the plain build."""
    _build()
    _build_asan()
    path, *_ = _scene(tmp_path)
    plain = subprocess.run([BIN, str(path)], capture_output=True, text=True, check=True)
    r = subprocess.run([BIN_ASAN, str(path)], capture_output=True, text=True, env=dict(os.environ, **SAN_ENV))
    assert r.returncode == 0, r.stderr[-2000:]
    _clean(r.stderr)
    assert r.stdout == plain.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_cpp_shim_matches_oracle(tmp_path, gpu_lib, monkeypatch, precision):
    # the C++ mirror leaves rbs_config.likelihood_precision open (DEFAULT): the environment picks it
    monkeypatch.setenv("RBS_PRECISION", precision)
    _build()
    path, o, default, deltas, frame, n = _scene(tmp_path)
    out = subprocess.run([BIN, str(path)], capture_output=True, text=True, check=True).stdout
    lines = {l.split()[0]: l.split()[1:] for l in out.strip().splitlines()}
    ll1 = np.array(lines["LL1"], dtype=np.float64)
    ll2 = np.array(lines["LL2"], dtype=np.float64)
    assert [int(x) for x in lines["IDX"]] == list(range(n))
    assert lines["ERR"] == ["ok"]
    poses = pose.compose_with_default(deltas, default, 2)
    o.reset()
    o.set_observation(frame)
    idx = np.zeros(n, np.int32)
    r1 = o.loglikes_poses(poses, idx, update=True)
    o.set_observation(frame)
    r2 = o.loglikes_poses(poses, np.arange(n - 1, -1, -1, dtype=np.int32), update=False)
    o.set_observation(frame)
    r3 = o.loglikes_poses(poses, np.arange(n - 1, -1, -1, dtype=np.int32), update=True)
    ll3 = np.array(lines["LL3"], dtype=np.float64)      # (the image borrowed: rbs_set_observation_borrowed, two-kernel launch)
    for got, ref in ((ll1, r1), (ll2, r2), (ll3, r3)):
        # F32: north_star's tolerance (these sums are well conditioned)
        assert (np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max() <= (1e-9 if precision == "f64" else 1e-5)
    if precision == "f64":
        # Parameters::occlusion_mode = "reference": three frames against the LAZY (reference-semantics) oracle at the transcendentals' accuracy
        om_ = ObjectModel([synth.mesh_m1(level=2)[0], synth.mesh_box12()[0]], [synth.mesh_m1(level=2)[1], synth.mesh_box12()[1]], center=True)
        lazy = ob.Oracle(om_, CameraData(synth.camera_matrix(80, 60), 60, 80), RbSensorBuilder.Parameters(sample_count=n), max_particles=n, mode=ob.LAZY)
        lazy.reset()
        idx = np.zeros(n, np.int32)
        for k in range(3):
            lazy.set_observation(frame)
            ref = lazy.loglikes_poses(poses, idx, update=True)
            got = np.array(lines[f"REF{k}"], dtype=np.float64)
            assert (np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max() <= 1e-10, k      # (the poses are composed on the device: an entry may differ from the host composition in its last bit)
            idx = np.arange(n - 1, -1, -1, dtype=np.int32)
        assert lines["ERR2"] == ["ok"]
    # the C++ tracker mirror against the Python device tracker: same device RNG key -> same states
    from dbot_ros_amd import RbSensor
    from dbot_ros_amd.tracker import DeviceParticleTracker, ObjectTransitionBuilder, ParticleTrackerBuilder
    om = ObjectModel([synth.mesh_m1(level=2)[0], synth.mesh_box12()[0]],
                     [synth.mesh_m1(level=2)[1], synth.mesh_box12()[1]], center=True)
    cam = CameraData(synth.camera_matrix(80, 60), 60, 80)
    P = RbSensorBuilder.Parameters(sample_count=n)
    with RbSensor(om, cam, P, max_particles=n) as s:
        tr = DeviceParticleTracker(ObjectTransitionBuilder(ObjectTransitionBuilder.Parameters(part_count=2)).build(),
                                   s, om, ParticleTrackerBuilder.Parameters(evaluation_count=n), device_rng=True, seed=42)
        init = default.copy().reshape(2, 12)
        for b in range(2):
            init[b, 0:3] -= pose.rotvec_to_matrix(init[b, 3:6]) @ om.centers[b]
        tr.initialize([init.ravel()])
        for k in range(3):
            est = tr.track(frame)
            got = np.array(lines[f"TRK{k}"], dtype=np.float64)
            assert np.abs(got - est).max() <= 1e-12, (k, np.abs(got - est).max())
            # ParticleTracker::submit / result with two frames in flight: the same estimates
            assert lines[f"PIP{k}"] == lines[f"TRK{k}"], k
        tr.close()


@pytest.mark.gpu
def test_native_host_bench_computes_what_python_computes(tmp_path, gpu_lib, monkeypatch):
    """tests/cpp/host_bench.cpp (bench.py's host_api_native_* leg: the host-pointer step driven from
    C++) on a small workload: the checksum of its last step's log-likelihoods equals the one the
    same calls give through ctypes."""
    import bench
    _build()
    exe = os.path.join(os.path.dirname(BIN), "host_bench")
    assert os.path.exists(exe)
    monkeypatch.setenv("RBS_PRECISION", "f32")
    from dbot_ros_amd import RbSensor
    n, rows, cols, F, steps, warm = 48, 120, 160, 5, 4, 3
    om = ObjectModel([synth.mesh_m1(level=2)[0]], [synth.mesh_m1(level=2)[1]], center=True)
    cam = CameraData(synth.camera_matrix(cols, rows), rows, cols)
    P = RbSensorBuilder.Parameters(sample_count=n)
    rng = np.random.default_rng(4)
    truths = [synth.truth_pose(1, frame=k) for k in range(F)]
    with RbSensor(om, cam, P, max_particles=n) as s:
        frames = np.stack([synth.make_frame(s.render_depth(t), rows, cols, rng) for t in truths]).astype(np.float32)
        poses = np.stack([synth.particle_poses(t, n, rng).reshape(n, -1) for t in truths])
        parents = rng.permutation(n).astype(np.int32)
        for i in range(warm + steps):
            s.set_observation(frames[i % F])
            ll = s.loglikes_poses(poses[i % F], parents.copy(), update=True)
    path = tmp_path / "w.bin"
    with open(path, "wb") as f:
        bench.write_host_workload(f, om, cam, P, frames, poses, parents, True)
    out = subprocess.run([exe, str(path), str(steps), str(warm)], capture_output=True, text=True, check=True).stdout
    tok = out.split()
    assert tok[0] == "host_bench", out
    want = float(ll[np.isfinite(ll)].sum())
    assert abs(float(tok[6]) - want) <= 1e-8 * max(1.0, abs(want)), (tok[6], want)
    # the same steps with each next frame handed over ahead (rbs_loglikes_prefetch / rbs_set_observation_prefetched): the same numbers
    out2 = subprocess.run([exe, "--prefetch", str(path), str(steps), str(warm)], capture_output=True, text=True, check=True).stdout
    assert out2.split()[6] == tok[6], (out2, out)
