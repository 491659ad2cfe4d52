import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
from dbot_ros_amd import CameraData, ObjectModel, RbSensor, RbSensorBuilder, synth
n=2000; v,f=synth.mesh_m1(); om=ObjectModel([v],[f]); cam=CameraData(synth.camera_matrix(),480,640)
prec=sys.argv[1] if len(sys.argv)>1 else None
with RbSensor(om,cam,RbSensorBuilder.Parameters(sample_count=n),max_particles=n,precision=prec) as s:
    rng=np.random.default_rng(0); truth=synth.truth_pose(1)
    frame=synth.make_frame(s.render_depth(truth),480,640,rng).astype(np.float32)
    poses=synth.particle_poses(truth,n,rng); idx=rng.permutation(n).astype(np.int32)
    for _ in range(5): s.set_observation(frame); s.loglikes_poses(poses, idx.copy(), update=True)
    t0=time.perf_counter(); K=50
    for _ in range(K):
        s.set_observation(frame); s.loglikes_poses(poses, idx.copy(), update=True)
    dt=(time.perf_counter()-t0)/K
    print("host-pointer API incl. frame upload (1.2 MB), poses H2D (192 KB), loglik D2H, sync: %.3f ms/call -> %.2f M particle-likelihoods/s"%(dt*1e3, n/dt/1e6))
