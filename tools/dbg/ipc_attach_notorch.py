"""Diagnostic: rbs_ipc_attach between two plain processes (no torch in either)."""
import faulthandler, os, sys, time, multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def worker(rank, conn, n, slab, with_torch):
    faulthandler.dump_traceback_later(25, exit=True)
    if with_torch:
        import torch; torch.cuda.set_device(0); torch.zeros(1, device="cuda")
    import scenarios as sc
    from dbot_ros_amd import RbSensor
    om, cam, P = sc.make_scene(("m1_l2",), 640, 480, max_particles=n)
    s = RbSensor(om, cam, P, max_particles=n, slab_px=slab)
    s.reset(); s.synchronize()
    conn.send(s.ipc_export()); other = conn.recv()
    blobs = [None, None]; blobs[rank] = s.ipc_export(); blobs[1 - rank] = other
    t = time.time()
    if rank == 0:
        s.ipc_attach(rank, blobs)
        print(f"n={n} slab={slab} torch={with_torch}: attach {time.time()-t:.2f}s", flush=True)
    conn.send(b"x"); conn.recv()
    s.close()

if __name__ == "__main__":
    n, slab, wt = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    ctx = mp.get_context("spawn")
    a, b = ctx.Pipe()
    ps = [ctx.Process(target=worker, args=(0, a, n, slab, wt)), ctx.Process(target=worker, args=(1, b, n, slab, wt))]
    [p.start() for p in ps]; [p.join() for p in ps]
