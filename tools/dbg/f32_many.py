import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import oracle_binding as ob, scenarios as sc
from dbot_ros_amd import RbSensor, synth
cols, rows = 1280, 960
for n in (24, 200, 1000, 6250):
    om, cam, P = sc.make_scene(("m4",), cols, rows, max_particles=n)
    render = ob.Oracle(om, cam, P, max_particles=1, mode=ob.EAGER)
    rng = np.random.default_rng(21)
    truth = synth.truth_pose(1)
    frame = synth.make_frame(render.render_depth(truth), rows, cols, rng)
    render.close()
    poses = synth.particle_poses(truth, n, rng, scale=2.0)
    for prec in ("f64", "f32"):
        with RbSensor(om, cam, P, max_particles=n, precision=prec) as g:
            g.reset(); g.set_observation(frame)
            ll = g.loglikes_poses(poses, np.zeros(n, np.int32), update=True)
            ll2 = g.loglikes_poses(poses, np.arange(n, dtype=np.int32), update=True)
            print(n, prec, "nan:", int(np.isnan(ll).sum()), int(np.isnan(ll2).sum()), "first", ll[:3], flush=True)
