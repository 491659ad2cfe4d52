"""Diagnostic: how long rbs_ipc_attach takes by handle size, and whether a process can attach twice in a row."""
import faulthandler, os, sys, time
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenarios as sc
from dbot_ros_amd import RbSensor
from dbot_ros_amd import dist as rdist

def worker(rank, port, cases):
    faulthandler.dump_traceback_later(25, exit=True)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=2)
    torch.cuda.set_device(0)
    for n, slab in cases:
        om, cam, P = sc.make_scene(("m1_l2",), 640, 480, max_particles=n)
        t0 = time.time()
        s = RbSensor(om, cam, P, max_particles=n, slab_px=slab)
        s.reset(); s.synchronize()
        t1 = time.time()
        rdist.attach_peers(s)
        t2 = time.time()
        dist.barrier()
        s.close()
        t3 = time.time()
        if rank == 0:
            print(f"n={n} slab={slab}: create {t1-t0:.2f}s attach {t2-t1:.2f}s close {t3-t2:.2f}s", flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    cases = [tuple(int(x) for x in c.split(':')) for c in sys.argv[2].split(',')]
    mp.spawn(worker, args=(int(sys.argv[1]), cases), nprocs=2)
