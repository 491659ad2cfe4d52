"""Spawned by tests/test_gpu_round6.py: the shared trail across PROCESSES (rbs_shared_trail_rebase, dist.PeerShardedStep
shared_trail=True).  Two ranks (gloo rendezvous) on cuda:0, handles attached over HIP IPC, an object that travels across the image;
the run is made twice -- planes against the scalar background, and against the shared plane (re-based every 4th step on global slot 0,
by every rank before the same step) -- and must agree bit for bit: every gathered log-likelihood, every parent, the planes themselves;
the stored windows must be smaller."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import scenarios as sc  # noqa: E402
from dbot_ros_amd import RbSensor, synth  # noqa: E402
from dbot_ros_amd import dist as rdist  # noqa: E402

PN, STEPS, WORLD, TEMP = 64, 22, 2, 4.0
COLS, ROWS = 320, 240


def inputs():
    n_all = PN * WORLD
    om, cam, P = sc.make_scene(("m1_l2",), COLS, ROWS, max_particles=2 * n_all)
    with RbSensor(om, cam, P, max_particles=1) as s:
        rng = np.random.default_rng(0)
        frames = []
        for k in range(STEPS):
            t = synth.truth_pose(1, frame=k)
            t[:, 9] += -0.12 + 0.012 * k          # 1.2 cm per frame across the image: a trail
            t[:, 10] += -0.05 + 0.005 * k
            frames.append((t, synth.make_frame(s.render_depth(t), ROWS, COLS, rng)))
    rng = np.random.default_rng(12)
    poses = [synth.particle_poses(t, n_all, rng, scale=1.0).reshape(n_all, -1) for t, _ in frames]
    g = torch.Generator().manual_seed(5)
    uniforms = [torch.rand(n_all, dtype=torch.float64, generator=g).sort().values for _ in frames]
    return om, cam, P, frames, poses, uniforms


def worker(rank, world, port, slab_px, occlusion, shared, q, precision=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        om, cam, P, frames, poses, uniforms = inputs()
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        n, cap = PN, 2 * PN
        with RbSensor(om, cam, P, max_particles=cap, slab_px=slab_px, occlusion=occlusion, precision=precision) as s:
            s.reset()
            rdist.attach_peers(s)

            def all_gather(out, inp):
                parts = [torch.empty(n, dtype=torch.float64) for _ in range(world)]
                dist.all_gather(parts, inp.cpu())
                out.copy_(torch.cat(parts))

            step = rdist.PeerShardedStep(s, n, cap, device=dev, min_share=2, stream=stream.cuda_stream, all_gather=all_gather,
                                         temperature=TEMP, fused=True, shared_trail=shared, trail_every=4, trail_threshold=-1.0)
            res = []
            for k, (_, frame) in enumerate(frames):
                s.set_observation(frame)
                d_poses = torch.from_numpy(poses[k][rank * n:(rank + 1) * n].copy()).to(dev)
                ps = step.step(d_poses, uniforms[k].to(dev))
                res.append((step.d_all.cpu().numpy().copy(), ps.cpu().numpy().copy()))
            torch.cuda.synchronize()
            dist.barrier()
            planes = [s.get_occlusion(q_) for q_ in range(0, n, 9)]
            area = float(np.mean([max(0, w[2] - w[0]) * max(0, w[3] - w[1]) for w in (s.get_window(q_) for q_ in range(n))]))
            state = s.shared_trail_state()
            dist.barrier()     # nobody unmaps while a peer may still be reading
        q.put((rank, res, planes, area, state))
    finally:
        dist.destroy_process_group()


def run(port, slab_px, occlusion, shared, precision=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, WORLD, port, slab_px, occlusion, shared, q, precision)) for r in range(WORLD)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in procs], key=lambda g: g[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0, p.exitcode
    return got


def main():
    port = int(sys.argv[1])
    # (attached ranks: slabs do not grow -- sized for the trail; the last case: the float32 likelihood)
    for slab_px, occlusion, precision in ((0, "device", None), (49152, "device", None), (0, "reference", None), (0, "device", "f32")):
        plain = run(port, slab_px, occlusion, False, precision)
        trail = run(port + 3, slab_px, occlusion, True, precision)
        port += 7
        for a, b in zip(plain, trail):
            assert a[4] == (False, 0) and b[4][0] and b[4][1] >= 4, (a[4], b[4])
            for k, ((lla, psa), (llb, psb)) in enumerate(zip(a[1], b[1])):
                assert np.array_equal(lla, llb), (k, float(np.abs(lla - llb).max()))
                assert np.array_equal(psa, psb), k
            for pa, pb in zip(a[2], b[2]):
                assert np.array_equal(pa, pb)
        print(f"slab_px={slab_px} occlusion={occlusion} precision={precision or 'f64'}: mean stored window, ranks 0/1: scalar background {plain[0][3]:.0f}/{plain[1][3]:.0f} px, "
              f"shared trail {trail[0][3]:.0f}/{trail[1][3]:.0f} px; re-basings {trail[0][4][1]}")
        assert trail[0][3] < 0.8 * plain[0][3] and trail[1][3] < 0.8 * plain[1][3]
    print("TRAIL_OK")


if __name__ == "__main__":
    main()
