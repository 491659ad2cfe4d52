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


@pytest.mark.parametrize("world,n,min_share,temp,spread", [
    (2, 48, 2, 1.0, 3.0),          # a handful of survivors
    (4, 500, 2, 1.0, 0.05),        # many distinct parents, most children alone with theirs
    (8, 2000, 2, 50.0, 30.0),      # BASELINE C1 over eight ranks, flattened weights
    (3, 1111, 3, 1.0, 1.0),        # chunks that do not divide evenly, min_share 3
    (2, 25000, 2, 1.0, 200.0),     # one survivor: every child of the other rank shares a remote parent
    (1, 64, 2, 1.0, 1.0),          # a single rank: nothing is remote
    (4, 3000, 2, 1.0, 6.0),        # some dozens of survivors, long weightless stretches between them
    (8, 25000, 2, 1.0, 2.0),       # BASELINE C3 over eight ranks
    (2, 5000, 3, 1.0, 4.0),        # more children than the plan keeps in LDS, min_share 3: the general plan in memory
    (4, 6250, 2, 1.0, 40.0),       # ... min_share 2: the one-sweep plan, few survivors
])
def test_peer_resample_matches_tensor_arithmetic(gpu_lib, world, n, min_share, temp, spread):
    """rbs_peer_resample (one call) against dist.global_resample + dist.plan_shard (the same step as tensor
    arithmetic, here on the CPU): identical parents, identical plan and counts, for every rank of the job."""
    from dbot_ros_amd import dist as rdist
    N = world * n
    rng = np.random.default_rng(world * 1000 + n)
    ll = rng.normal(-3000.0, spread, N)
    ll[rng.integers(0, N, 3)] = np.nan if n > 100 else ll[0]     # contained particles weigh nothing
    g = torch.Generator().manual_seed(n)
    u_sorted = torch.sort(torch.rand(N, dtype=torch.float64, generator=g)).values
    ll_ref = torch.from_numpy(np.where(np.isnan(ll), -np.inf, ll))
    ps = rdist.global_resample(ll_ref, u_sorted, temp)
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=2 * n)
    with RbSensor(om, cam, P, max_particles=2 * n) as s:
        d_ll = torch.from_numpy(ll).cuda()
        d_u = u_sorted.cuda()
        for rank in range(world):
            out = [torch.full((n,), -7, dtype=torch.int32, device="cuda") for _ in range(4)]
            counts = torch.zeros(4, dtype=torch.int64, device="cuda")
            for _ in range(2):      # twice: the counts accumulate
                s.peer_resample(d_ll.data_ptr(), d_u.data_ptr(), N, n, rank, min_share, temp, out[0].data_ptr(), out[1].data_ptr(),
                                out[2].data_ptr(), out[3].data_ptr(), counts.data_ptr())
            s.synchronize()
            torch.cuda.synchronize()
            pidx, src, dst, cnt = rdist.plan_shard(ps, n, 2 * n, rank, min_share)
            mine = ps[rank * n:(rank + 1) * n]
            assert np.array_equal(out[3].cpu().numpy(), mine.numpy()), rank
            assert np.array_equal(out[0].cpu().numpy(), pidx.numpy()), rank
            assert np.array_equal(out[1].cpu().numpy(), src.numpy()), rank
            assert np.array_equal(out[2].cpu().numpy(), dst.numpy()), rank
            runs = int((mine[1:] != mine[:-1]).sum()) + 1
            assert counts.cpu().tolist() == [2 * int(c) for c in cnt] + [2 * runs], rank
        with pytest.raises(RbSensorError):       # the staging slots must exist
            s.peer_resample(d_ll.data_ptr(), d_u.data_ptr(), N * 2, 2 * n, 0, 2, 1.0, out[0].data_ptr(), out[1].data_ptr(),
                            out[2].data_ptr(), 0, counts.data_ptr())
