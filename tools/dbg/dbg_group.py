import os, sys
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
import numpy as np, scenarios as sc
from dbot_ros_amd import RbSensor, synth
n, cols, rows = 48, 160, 120
om, cam, P = sc.make_scene(("m1_l2",), cols, rows, max_particles=n)
for layout in ("dense", "window"):
  for ids in ([0, 0], [0, 0, 0]):
    ref = RbSensor(om, cam, P, max_particles=n, precision="f64", state_layout="dense")
    g = RbSensor(om, cam, P, max_particles=n, precision="f64", state_layout=layout, device_ids=ids)
    rng = np.random.default_rng(1)
    idx = np.zeros(n, np.int32)
    bad = None
    for k in range(12):
        t = synth.truth_pose(1, frame=k)
        frame = synth.make_frame(ref.render_depth(t), rows, cols, rng)
        m = [48, 48, 30, 48, 20, 48, 40, 10, 48, 33, 48, 48][k]
        poses = synth.particle_poses(t, m, rng)
        par = idx[:m].copy()
        ref.set_observation(frame); g.set_observation(frame)
        lr = ref.loglikes_poses(poses, par.copy(), update=True)
        lg = g.loglikes_poses(poses, par.copy(), update=True)
        d = np.abs(lr - lg) / np.maximum(1, np.abs(lr))
        if d.max() > 1e-12 and bad is None:
            bad = (k, m, np.nonzero(d > 1e-12)[0][:10], par[np.nonzero(d > 1e-12)[0][:10]])
        idx = rng.integers(0, m, n).astype(np.int32)
    print(layout, ids, "first bad:", bad)
    ref.close(); g.close()
