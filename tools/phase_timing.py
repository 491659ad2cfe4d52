#!/usr/bin/env python3
"""Per-phase cycle breakdown of the raster kernel (needs a -DRBS_PHASE_TIMING build:
RBS_LIB_PATH=build_variants/phase.so python tools/phase_timing.py [--update 0|1])."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dbot_ros_amd import CameraData, ObjectModel, RbSensor, RbSensorBuilder, synth, _capi
upd = int(sys.argv[sys.argv.index("--update") + 1]) if "--update" in sys.argv else 0
names = sys.argv[sys.argv.index("--mesh") + 1].split(",") if "--mesh" in sys.argv else ["m1"]
n = int(sys.argv[sys.argv.index("--particles") + 1]) if "--particles" in sys.argv else 2000
cols = int(sys.argv[sys.argv.index("--cols") + 1]) if "--cols" in sys.argv else 640
rows = cols * 3 // 4
zdist = float(sys.argv[sys.argv.index("--z") + 1]) if "--z" in sys.argv else 0.7
fns = {"m1": synth.mesh_m1, "m2": synth.mesh_m2, "m3": synth.mesh_m3, "m4": synth.mesh_m4}
ms_ = [fns[k]() for k in names]
om = ObjectModel([v for v, _ in ms_], [t for _, t in ms_]); cam = CameraData(synth.camera_matrix(cols, rows), rows, cols)
P = RbSensorBuilder.Parameters(sample_count=n)
with RbSensor(om, cam, P, max_particles=n) as s:
    lib = _capi.load()
    rng = np.random.default_rng(0)
    truth = synth.truth_pose(len(names), z=zdist)
    s.set_observation(synth.make_frame(s.render_depth(truth), rows, cols, rng))
    poses = synth.particle_poses(truth, n, rng)
    idx = rng.permutation(n).astype(np.int32)
    out = (C.c_ulonglong * 32)()
    for rep in range(3):
        s.loglikes_poses(poses, idx.copy(), update=bool(upd))
        lib.rbs_debug_phase_cycles(s._h, out)
        c = np.array(list(out), dtype=np.float64)
        names = ["-", "clear", "raster(+barriers)", "pixel pass", "reduce"]
        tot = c[1:5].sum()
        print("rep", rep, "ms", "%.3f" % s.last_kernel_ms(), {names[k]: "%.1f%%" % (100 * c[k] / tot) for k in range(1, 5)},
              "wave0 cycles per item: %.0f" % (tot / n),
              "\n   raster phase of wave 0:", {nm: "%.1f%%" % (100 * c[k] / max(c[2], 1)) for k, nm in
                                           ((8, "cluster cull"), (9, "pre-test+queue"), (10, "before setup"), (11, "setup"), (12, "sample loops"),
                                            (13, "barrier wait"), (14, "big triangles"))},
              "\n   per item, cycles: ticket %.0f, descriptor %.0f, pose + eye %.0f, cluster cull %.0f" % (c[7] / n, c[15] / n, c[0] / n, c[8] / n),
              "\n   per item (wave 0 of four; clusters are tested by every wave): clusters tested %.1f, surviving the cull %.1f (%.0f %%), "
              "setup batches of wave 0 %.2f (of them whole clusters with shared vertices %.2f), triangles of its last partial batches %.1f, "
              "triangles handed to the block-wide path %.2f" % (c[16] / max(c[21], 1), c[17] / max(c[21], 1), 100 * c[17] / max(c[16], 1), c[18] / max(c[21], 1),
                                                             c[19] / max(c[21], 1), c[23] / max(c[21], 1), c[20] / max(c[21], 1)),
              "\n   items per particle %.2f" % (c[21] / n),
              "\n   | block lifetime: %.0f cycles, %.1f us wall -> %.2f GHz shader clock" % (c[5] / 768, c[6] / 768 / 100.0, c[5] / max(c[6], 1) * 0.1))
