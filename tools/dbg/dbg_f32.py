import sys; import os; R=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
import numpy as np, oracle_binding as ob, scenarios as sc
from dbot_ros_amd import RbSensor, synth
n=24
om,cam,P=sc.make_scene(("m1",),640,480,max_particles=n)
eager=ob.Oracle(om,cam,P,max_particles=n,mode=ob.EAGER)
frames=sc.make_frames(eager,1,4,seed=3)
with RbSensor(om,cam,P,max_particles=n,precision="f32") as g, RbSensor(om,cam,P,max_particles=n,precision="f64") as g64:
    a=sc.run_sequence(g,frames,n); b=sc.run_sequence(g64,frames,n)
    for k in range(4):
        d=a[k]-b[k]
        i=np.argmax(np.abs(d))
        print(k,'max abs',np.abs(d).max(),'at ll',b[k][i],'mean abs',np.abs(d).mean(),'mean signed',d.mean(),'|ll| median',np.median(np.abs(b[k])))
    for slot in range(3):
        pa,pb=g.get_occlusion(slot),g64.get_occlusion(slot)
        dd=np.abs(pa-pb); print('plane',slot,'max',dd.max(),'n>1e-6',(dd>1e-6).sum(),'n diff',(dd>0).sum())
        j=np.argmax(dd); print('   worst vals',pa[j],pb[j])
