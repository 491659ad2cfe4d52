#!/usr/bin/env python3
"""bench.py -- particle-likelihoods/s of the RbSensor hot path on MI355X.

A "step" is one frame of BASELINE.json config C1 on one GPU: set_observation(frame k) +
RbSensor::loglikes(update=true) over 2 000 particles, one 5 120-triangle mesh (M1), 640x480
synthetic frames of SURVEY 8d's sequence (the object translates 2 mm and turns 1 degree per
frame; 30 frames played forwards and backwards), every child inheriting from a distinct random
parent slot (permutation: no occlusion plane is read twice, the worst case for HBM traffic).
Inputs (frames, poses, parent indices, occlusion planes) are resident in HBM before the timed
region.  --sequence 0 replays one frame and one pose set for ever (a resting object).  With --gpus N each
rank evaluates its own 2 000-particle shard (weak scaling) and the per-particle
log-likelihoods are all-gathered over RCCL every step (the weight exchange before resampling).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline      algorithmic bytes (2*4*W*H per particle-likelihood) / live kernel time
  cpu_baseline  the CPU oracle (reference CPU-path semantics) timed on this host, 1 thread
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles", type=int, default=2000, help="particles per GPU")
    ap.add_argument("--cols", type=int, default=640)
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--mesh", default="m1")
    ap.add_argument("--parents", default="permutation",
                    choices=["permutation", "identity", "resampled", "peaked", "peaked_unsorted"])
    ap.add_argument("--update", type=int, default=1, help="0: read-only evaluation (non-final blocks)")
    ap.add_argument("--sequence", type=int, default=30, help="frames in the moving-object sequence (0: one static frame)")
    ap.add_argument("--config", default=None, choices=["c1", "c1_readonly", "c2", "c3_slice", "c4_slice", "default_res"],
                    help="a row of BASELINE.md section 3 (sets mesh / particles / resolution / steps; c1 = no flags)")
    ap.add_argument("--fill-planes", type=float, default=None,
                    help="start from planes that differ from the background everywhere (windows = whole frame): "
                         "the windowed layout's worst case")
    ap.add_argument("--fill-fraction", type=float, default=1.0,
                    help="with --fill-planes: only a central rectangle of this fraction of the frame")
    ap.add_argument("--precision", default=None, choices=["f64", "f32"], help="likelihood precision (default: the library's)")
    ap.add_argument("--no-dense-leg", action="store_true", help="skip the whole-plane (RBS_STATE=dense) comparison run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    a = ap.parse_args()
    presets = {"c1": {}, "c1_readonly": {"update": 0},
               "c2": {"mesh": "m1,m2,m3", "particles": 6666, "steps": 50},
               "c3_slice": {"particles": 25000, "steps": 20, "warmup": 3},
               "c4_slice": {"mesh": "m4", "cols": 1280, "rows": 960, "particles": 6250, "steps": 5, "warmup": 2},
               "default_res": {"cols": 80, "rows": 60}}
    for k, v in presets.get(a.config, {}).items():
        setattr(a, k, v)
    return a


def cpu_baseline(om, cam, P, truth, frame, seconds):
    """The oracle in reference-CPU-semantics mode (LAZY) on a bounded sample of the same
    workload: config C0's 200 particles per call, repeated frame after frame (update=true,
    permuted parents) until `seconds` elapse.  Headline = single thread, as dbot's CPU model
    runs; `all_cores` = the same loop with OpenMP over particles on every host core."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    from dbot_ros_amd import synth
    n = 200
    tris = sum(len(t) for t in om.triangles)

    def run(threads, budget):
        orc = ob.Oracle(om, cam, P, max_particles=n, mode=ob.LAZY)
        rng = np.random.default_rng(1)
        poses = synth.particle_poses(truth, n, rng)
        idx = np.zeros(n, dtype=np.int32)
        orc.reset()
        done, t0 = 0, time.perf_counter()
        while True:
            orc.set_observation(frame)
            orc.loglikes_poses(poses, idx, update=True, threads=threads)
            done += n
            idx = rng.permutation(n).astype(np.int32)
            el = time.perf_counter() - t0
            if el >= budget:
                break
        orc.close()
        return done, el

    done, el = run(1, seconds)
    cores = os.cpu_count() or 1
    threads = min(cores, n)
    done_mt, el_mt = run(threads, max(3.0, seconds / 3))
    model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": done / el, "unit": "particle-likelihoods/s", "cores": 1, "kind": "port", "host_cpu": model, "host_cores": cores,
            "sample": f"{done} particle-likelihoods = {done // n} loglikes(update=true) calls x {n} "
                      f"particles, {cam.cols}x{cam.rows}, {tris} triangles, {el:.1f} s on 1 of {cores} host cores",
            "all_cores": {"value": done_mt / el_mt, "cores": threads,
                          "sample": f"{done_mt} particle-likelihoods in {el_mt:.1f} s, OpenMP over particles"}}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path is the only path (no CPU fallback)")
    # RBS_BENCH_BACKEND=gloo: functional test of the multi-rank path on a box with fewer GPUs than
    # ranks (ranks share devices, the all-gather goes through host tensors); never used for numbers
    backend = os.environ.get("RBS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from dbot_ros_amd import CameraData, ObjectModel, RbSensor, RbSensorBuilder, synth
    mesh_fns = {"m1": synth.mesh_m1, "m2": synth.mesh_m2, "m3": synth.mesh_m3, "m4": synth.mesh_m4,
                "box12": synth.mesh_box12}
    meshes = [mesh_fns[m]() for m in a.mesh.split(",")]
    nb = len(meshes)
    f = np.concatenate([t for _, t in meshes])
    om = ObjectModel([v for v, _ in meshes], [t for _, t in meshes], center=True)
    cam = CameraData(synth.camera_matrix(a.cols, a.rows), a.rows, a.cols)
    n = a.particles
    P = RbSensorBuilder.Parameters(sample_count=n)
    sensor = RbSensor(om, cam, P, device_id=local, max_particles=n, precision=a.precision)

    # synthetic frame: the product's own render hook supplies the object's depth
    rng = np.random.default_rng(0)
    prng = np.random.default_rng(1 + rank)
    F = max(1, a.sequence)
    truths = [synth.truth_pose(nb, frame=k) for k in range(F)]
    frames = np.stack([synth.make_frame(sensor.render_depth(t), a.rows, a.cols, rng) for t in truths])
    poses_seq = np.stack([synth.particle_poses(t, n, prng).reshape(n, -1) for t in truths])
    order = list(range(F)) + list(range(F - 2, 0, -1))      # forwards, then backwards
    truth, frame = truths[0], frames[0]
    if a.parents == "permutation":
        parents = synth.resample_like_indices(n, prng)
    elif a.parents == "identity":
        parents = np.arange(n, dtype=np.int32)
    elif a.parents == "resampled":
        parents = synth.resample_like_indices(n, prng, concentration=1.0)
    else:  # a tracker's usual regime: few survivors, many siblings
        parents = synth.resample_like_indices(n, prng, concentration=0.02)
        if a.parents == "peaked_unsorted":
            parents = prng.permutation(parents).astype(np.int32)
    if rank == 0:
        print(f"# parents={a.parents}: {len(np.unique(parents))} distinct of {n}", file=sys.stderr)

    dev = torch.device("cuda", local)
    d_poses = torch.from_numpy(poses_seq).to(dev)                       # [F][n][12*bodies]
    d_frames = torch.from_numpy(frames.astype(np.float32)).to(dev)     # [F][rows*cols]
    d_idx = torch.from_numpy(parents).to(dev)
    d_out = torch.empty(n, dtype=torch.float64, device=dev)
    d_all = torch.empty(n * world, dtype=torch.float64, device=dev) if world > 1 else None
    sensor.reset()
    if a.fill_planes is not None:
        plane = np.full((a.rows, a.cols), np.float32(0.1), dtype=np.float32)      # = the background after reset
        fr = float(np.sqrt(min(1.0, max(0.0, a.fill_fraction))))
        r0, c0 = int(a.rows * (1 - fr) / 2), int(a.cols * (1 - fr) / 2)
        plane[r0:a.rows - r0, c0:a.cols - c0] = a.fill_planes
        plane = plane.ravel()
        for slot in range(n):
            sensor.set_occlusion(slot, plane)
    sensor.set_observation(frame)
    sensor.synchronize()
    # a non-default torch stream: the kernel, the timing events and the RCCL all-gather all
    # live on it (a NULL stream would select the handle's private stream instead)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    d_out.zero_()                  # first submission creates the stream's hardware queue: setup, not a step
    torch.cuda.synchronize()

    def exchange():
        # the weight exchange before resampling: every rank gets all N*world log-likelihoods
        if backend == "nccl":
            dist.all_gather_into_tensor(d_all, d_out)      # RCCL over xGMI, on `stream`
        else:
            host = [torch.empty(n, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(host, d_out.cpu())
            d_all.copy_(torch.cat(host))

    pose_bytes = d_poses[0].numel() * 8
    frame_bytes = d_frames[0].numel() * 4
    counter = [0]

    def launch(sn):
        k = order[counter[0] % len(order)]
        counter[0] += 1
        if a.sequence > 0:
            sn.set_observation_device(d_frames.data_ptr() + k * frame_bytes, stream.cuda_stream)
        sn.loglikes_device(d_poses.data_ptr() + k * pose_bytes, d_idx.data_ptr(), n, bool(a.update),
                           d_out.data_ptr(), stream.cuda_stream)

    def step():
        launch(sensor)
        if world > 1:
            exchange()

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # events are packets the stream retires between kernels: bracket every 4th step only (the
    # library samples its own kernel timing events the same way)
    k_ev = []
    t0 = time.perf_counter()
    for i in range(a.steps):
        if i % 4 == 0:
            k_ev.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
            k_ev[-1][0].record(stream)
        launch(sensor)
        if i % 4 == 0:
            k_ev[-1][1].record(stream)
        if world > 1:
            exchange()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank must hold every rank's log-likelihoods after the exchange
        ref = d_all.view(world, n)[rank].cpu().numpy()
        if not np.array_equal(ref, d_out.cpu().numpy()):
            raise SystemExit("all-gather returned the wrong shard")
    call_ms = float(np.mean([s.elapsed_time(e) for s, e in k_ev]))
    # the dominant kernel (rbs_copy_kernel) runs on the library's second stream: its duration
    # comes from the HIP events the library records on THAT stream, averaged over the timed steps
    lib_call_ms, copy_ms, n_used = sensor.timing_summary(a.steps)
    raster_ms = sensor.raster_kernel_ms(a.steps)
    # the dominant kernel: the raster kernel on windowed planes, the copy kernel on whole planes
    kernel_name, kernel_ms = ("rbs_copy_kernel", copy_ms) if (a.update and copy_ms > raster_ms) else ("rbs_raster_kernel", raster_ms)
    windows = np.array([sensor.get_window(s_) for s_ in range(0, n, max(1, n // 64))])
    win_frac = float(np.mean(np.maximum(0, windows[:, 2] - windows[:, 0]) * np.maximum(0, windows[:, 3] - windows[:, 1]))) / (a.rows * a.cols)
    ll = d_out.cpu().numpy()
    if not np.isfinite(ll).all():
        raise SystemExit("non-finite log-likelihoods in the timed run")

    if rank == 0:
        alg_bytes = (2.0 if a.update else 1.0) * 4.0 * a.rows * a.cols * n  # per launch, SURVEY 8d
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        layout = "dense" if os.environ.get("RBS_STATE") == "dense" else "window"
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("state_layout", "dense") == layout:
                    traffic = j.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # what bounds the dominant kernel when it is not HBM: VALU issue, from the committed SQ
        # counter summary of this same command (tools/profile_round.sh); live: measured HBM rate
        valu = None
        sqf = os.path.join(ROOT, "profiles", "r01_raster_sq.json" if layout == "window" else "r01_dense_raster_sq.json")
        if os.path.exists(sqf) and kernel_name == "rbs_raster_kernel":
            try:
                j = json.load(open(sqf))
                per = j["per_dispatch"]
                valu = {"valu_wave_instructions_per_launch": per["SQ_INSTS_VALU"],
                        "issue_rate_G_per_s": per["SQ_INSTS_VALU"] / (kernel_ms * 1e-3) / 1e9,
                        "peak_issue_rate_G_per_s": 256 * 4 * 2.4 / 4.0 * 1.0,   # 1 024 SIMDs, one wave64 VALU op per 4 cycles, 2.4 GHz
                        "source": os.path.relpath(sqf, ROOT)}
                valu["frac"] = valu["issue_rate_G_per_s"] / valu["peak_issue_rate_G_per_s"]
            except Exception:
                valu = None
        out = {
            "metric": "particle-likelihoods/sec @640x480" if (a.cols, a.rows) == (640, 480)
                      else f"particle-likelihoods/sec @{a.cols}x{a.rows}",
            "value": n * world * a.steps / elapsed,
            "unit": "particle-likelihoods/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C1: {n} particles/GPU x loglikes(update={'true' if a.update else 'false'}), {a.cols}x{a.rows} "
                                   f"synthetic depth frame, mesh {a.mesh} ({len(f)} triangles), "
                                   f"parents={a.parents}, " + (f"{F}-frame moving-object sequence" if a.sequence > 0 else "one static frame"),
                       "particles_per_gpu": n, "resolution": [a.cols, a.rows], "triangles": int(len(f)),
                       "sharding": f"particles/{world}" + (" + RCCL all-gather of log-likelihoods" if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": kernel_name,
                         "kernel_ms": kernel_ms, "kernel_launches_averaged": n_used,
                         "raster_kernel_ms": raster_ms, "copy_kernel_ms": copy_ms,
                         "state_layout": layout, "stored_window_fraction_of_plane": win_frac,
                         "measured_hbm_GBps": (traffic / (kernel_ms * 1e-3) / 1e9) if traffic else None,
                         "valu_issue": valu,
                         "call_ms_launch_stream": call_ms,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and a.update and layout == "window" and not a.no_dense_leg:
            # the same steps on whole planes (every updating call copies every plane in full):
            # what the windowed layout falls back to when windows grow to the whole frame
            os.environ["RBS_STATE"] = "dense"
            dense = RbSensor(om, cam, P, device_id=local, max_particles=n)
            os.environ.pop("RBS_STATE")
            dense.reset()
            dense.set_observation(frame)
            dense.synchronize()
            counter[0] = 0
            for _ in range(a.warmup):
                launch(dense)
            torch.cuda.synchronize()
            dsteps = min(a.steps, 120)
            td = time.perf_counter()
            for _ in range(dsteps):
                launch(dense)
            torch.cuda.synchronize()
            td = time.perf_counter() - td
            _, dcopy_ms, _ = dense.timing_summary(dsteps)
            out["dense_state"] = {"value": n * dsteps / td, "unit": "particle-likelihoods/s", "steps": dsteps,
                                  "ms_per_step": td / dsteps * 1e3,
                                  "roofline": {"bound": "hbm", "kernel": "rbs_copy_rows_kernel", "kernel_ms": dcopy_ms,
                                               "achieved": alg_bytes / (dcopy_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                                               "unit": "GB/s", "frac": alg_bytes / (dcopy_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}}
            dense.close()
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(om, cam, P, truth, frame, a.cpu_seconds)
        print(json.dumps(out), flush=True)
    sensor.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
