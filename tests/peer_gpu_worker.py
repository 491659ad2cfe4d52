"""Spawned by tests/test_gpu_peers.py: two ranks (gloo rendezvous), BOTH on cuda:0, each with its own handle attached
to the other's through rbs_ipc_export / rbs_ipc_attach; dist.PeerShardedStep with min_share = 2, so remote parents
are read in place through the mapped buffers AND (shared ones) pulled into staging slots by rbs_stage_windows.
Rank 0 then replays the run on ONE handle holding all particles: log-likelihoods must agree bit for bit."""
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

PN, STEPS, WORLD, TEMP = 48, 6, 2, 60.0


def inputs():
    n_all = PN * WORLD
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=2 * n_all)
    with RbSensor(om, cam, P, max_particles=1) as s:
        rng = np.random.default_rng(0)
        frames = []
        for k in range(STEPS):
            t = synth.truth_pose(1, frame=k)
            frames.append((t, synth.make_frame(s.render_depth(t), 120, 160, rng)))
    rng = np.random.default_rng(12)
    poses = [synth.particle_poses(t, n_all, rng, scale=2.0).reshape(n_all, -1) for t, _ in frames]   # by SLOT
    g = torch.Generator().manual_seed(5)
    uniforms = [torch.rand(n_all, dtype=torch.float64, generator=g) for _ in frames]
    return om, cam, P, frames, poses, uniforms


def single(slab_px):
    om, cam, P, frames, poses, uniforms = inputs()
    n_all = PN * WORLD
    out = []
    with RbSensor(om, cam, P, max_particles=n_all, slab_px=slab_px) as s:
        s.reset()
        idx = np.zeros(n_all, np.int32)
        for k, (_, frame) in enumerate(frames):
            s.set_observation(frame)
            ll = s.loglikes_poses(poses[k], idx, update=True)
            ps = rdist.global_resample(torch.from_numpy(ll), uniforms[k], TEMP)
            out.append((ll.copy(), ps.numpy().copy()))
            idx = ps.numpy().astype(np.int32)
    return out


def worker(rank, world, port, slab_px, q, fused=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        om, cam, P, frames, poses, uniforms = inputs()
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        n, cap = PN, 2 * PN
        with RbSensor(om, cam, P, max_particles=cap, slab_px=slab_px) as s:
            s.reset()
            rdist.attach_peers(s)

            def all_gather(out, inp):   # (gloo: through the host; bench.py runs RCCL on the stream)
                parts = [torch.empty(n, dtype=torch.float64) for _ in range(world)]
                dist.all_gather(parts, inp.cpu())
                out.copy_(torch.cat(parts))

            step = rdist.PeerShardedStep(s, n, cap, device=dev, min_share=2, stream=stream.cuda_stream, all_gather=all_gather,
                                         temperature=TEMP, fused=fused)
            res = []
            for k, (_, frame) in enumerate(frames):
                s.set_observation(frame)
                d_poses = torch.from_numpy(poses[k][rank * n:(rank + 1) * n].copy()).to(dev)
                u = uniforms[k].sort().values if fused else uniforms[k]    # (fused: one launch, sorted uniforms; the same parents)
                ps = step.step(d_poses, u.to(dev))
                res.append((step.d_all.cpu().numpy().copy(), ps.cpu().numpy().copy()))
            torch.cuda.synchronize()
            dist.barrier()     # nobody unmaps while a peer may still be reading
            counts = step.counts.cpu().tolist()
        q.put((rank, res, counts))
    finally:
        dist.destroy_process_group()


def main():
    port = int(sys.argv[1])
    for slab_px, fused in ((0, False), (4096, False), (0, True), (4096, True)):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=worker, args=(r, WORLD, port, slab_px, q, fused)) for r in range(WORLD)]
        for p in procs:
            p.start()
        got = [q.get(timeout=300) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0, p.exitcode
        ref = single(slab_px)
        remote = sum(g[2][0] for g in got)
        from_staging = sum(g[2][1] for g in got)
        staged = sum(g[2][2] for g in got)
        print(f"slab_px={slab_px} fused={fused}: children with a remote parent {remote}, of them served from staging {from_staging}, planes staged {staged}")
        assert remote > from_staging > 0 and staged > 0, "the scenario must exercise in-place remote reads AND staging"
        for rk, res, _ in got:
            for k, ((ll_ref, ps_ref), (ll, ps)) in enumerate(zip(ref, res)):
                assert np.array_equal(ps, ps_ref[rk * PN:(rk + 1) * PN] if fused else ps_ref), k
                assert np.array_equal(ll, ll_ref), (k, np.abs(ll - ll_ref).max())
        port += 7
    print("PEERS_OK")


if __name__ == "__main__":
    main()
