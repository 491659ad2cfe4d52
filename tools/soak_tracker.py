import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
from dbot_ros_amd import CameraData, ObjectModel, RbSensor, RbSensorBuilder, pose, synth
from dbot_ros_amd.tracker import DeviceParticleTracker, ObjectTransitionBuilder, ParticleTrackerBuilder
n=2000; cols,rows=640,480
v,f=synth.mesh_m3(); om=ObjectModel([v],[f]); cam=CameraData(synth.camera_matrix(cols,rows),rows,cols)
P=RbSensorBuilder.Parameters(sample_count=n)
with RbSensor(om,cam,P,max_particles=n) as s:
    tr=DeviceParticleTracker(ObjectTransitionBuilder(ObjectTransitionBuilder.Parameters()).build(), s, om, ParticleTrackerBuilder.Parameters(evaluation_count=n), device_rng=True, seed=5)
    def truth_state(k):
        Rt=synth.truth_pose(1,frame=k)[0]; st=np.zeros(12); st[3:6]=pose.matrix_to_rotvec(Rt[:9].reshape(3,3)); st[0:3]=Rt[9:]-Rt[:9].reshape(3,3)@om.centers[0]; return st
    tr.initialize([truth_state(0)])
    rng=np.random.default_rng(0); errs=[]; t0=time.time()
    for k in range(1,601):
        kk = k if (k//60)%2==0 else 60-(k%60) + (k//60)*0   # back and forth motion within 60 frames
        kk = k%120 if k%120<60 else 120-(k%120)
        fr=synth.make_frame(s.render_depth(synth.truth_pose(1,frame=kk)),rows,cols,rng,occluder=(k%200>150))
        est=tr.track(fr); errs.append(np.linalg.norm(est[0:3]-truth_state(kk)[0:3]))
        assert np.isfinite(est).all()
    errs=np.array(errs); print("frames 600 time %.1fs max err %.4f mean %.4f last100 max %.4f resamplings %d"%(time.time()-t0, errs.max(), errs.mean(), errs[-100:].max(), tr.n_resamplings))
    occ=s.get_occlusion(0); print("occ plane range", occ.min(), occ.max(), np.isnan(occ).sum())
    tr.close()
