"""One process per GPU with the ranks' handles attached to each other (rbs_ipc_export / rbs_ipc_attach, round 4),
and the window-sized plane transport (rbs_export_window / rbs_import_window / rbs_stage_windows)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import scenarios as sc
from dbot_ros_amd import RbSensor, synth
from dbot_ros_amd.sensor import RbSensorError

pytestmark = pytest.mark.gpu


def _run_some_frames(s, n, frames=3, seed=3):
    rng = np.random.default_rng(seed)
    s.reset()
    idx = np.zeros(n, np.int32)
    for k in range(frames):
        truth = synth.truth_pose(1, frame=k)
        s.set_observation(synth.make_frame(s.render_depth(truth), s.rows, s.cols, rng))
        s.loglikes_poses(synth.particle_poses(truth, n, rng, scale=2.0), idx, update=True)
        idx = rng.integers(0, n, n).astype(np.int32)


@pytest.mark.parametrize("slab_px,layout", [(0, None), (8192, None), (0, "dense")])
def test_window_export_import_and_staging_copy_planes_exactly(gpu_lib, slab_px, layout):
    """A plane sent as (rectangle, w x h values) and stored into another slot is the same plane, in every state
    layout; rbs_stage_windows does the same on the device for a list of slots."""
    n = 8
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=2 * n)
    with RbSensor(om, cam, P, max_particles=2 * n, slab_px=slab_px, state_layout=layout) as s:
        _run_some_frames(s, n)
        buf = torch.empty(160 * 120, dtype=torch.float32, device="cuda")
        moved = 0
        for slot in range(n):
            rect = s.export_window(slot, buf.data_ptr(), buf.numel())
            x0, y0, x1, y1 = rect
            if layout != "dense":
                assert rect == s.get_window(slot) or (x1 <= x0 and s.get_window(slot)[2] <= s.get_window(slot)[0])
                assert (x1 - x0) * (y1 - y0) < 0.5 * 160 * 120          # a window, not the frame
            moved += max(0, x1 - x0) * max(0, y1 - y0)
            s.import_window(n + slot, rect, buf.data_ptr())
            s.synchronize()
            assert np.array_equal(s.get_occlusion(n + slot), s.get_occlusion(slot)), slot
        assert moved > 0
        # a buffer that is too small is refused, and says how large the window is
        with pytest.raises(RbSensorError):
            s.export_window(0, buf.data_ptr(), 4)
        if layout != "dense":
            src = torch.tensor([3, 1, 0, 0], dtype=torch.int32, device="cuda")
            dst = torch.tensor([n + 5, -1, n + 6, -1], dtype=torch.int32, device="cuda")
            s.stage_windows(src.data_ptr(), dst.data_ptr(), 4)
            s.synchronize()
            assert np.array_equal(s.get_occlusion(n + 5), s.get_occlusion(3))
            assert np.array_equal(s.get_occlusion(n + 6), s.get_occlusion(0))
            assert s.get_window(n + 5) == s.get_window(3)


def test_peer_step_two_ranks_on_one_gpu(gpu_lib):
    """tests/peer_gpu_worker.py: two processes on cuda:0, handles attached over HIP IPC, dist.PeerShardedStep --
    remote parents read in place through the mapped buffers, shared ones staged -- against one handle holding all
    particles: bit-identical log-likelihoods, identical global parents; whole planes and slabs."""
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "peer_gpu_worker.py")
    port = 28100 + os.getpid() % 1000
    r = subprocess.run([sys.executable, script, str(port)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "PEERS_OK" in r.stdout, r.stdout[-2000:]
    print(r.stdout[-600:])


def test_ipc_attach_refuses_mismatched_handles(gpu_lib):
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=8)
    with RbSensor(om, cam, P, max_particles=8) as a, RbSensor(om, cam, P, max_particles=16) as b:
        blobs = [a.ipc_export(), b.ipc_export()]
        with pytest.raises(RbSensorError) as e:
            a.ipc_attach(0, blobs)
        assert "differs" in str(e.value)
        with pytest.raises(RbSensorError):
            a.ipc_attach(0, [b"\0" * 512, b"\0" * 512])
