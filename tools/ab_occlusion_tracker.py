"""Device tracker frames/s in both occlusion modes (rbs_config.occlusion_mode), same sequence, same seeds: tools/ab_occlusion.sh's
companion for the workload a FILTER produces (resampled parents: shared, cache-resident planes) instead of bench.py's permutation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
a = bench.parse()
om, cam, P, n_tri, nb = bench.build_scene(a)
dev = torch.device("cuda", 0)
for rep in range(2):
    for occ in ("device", "reference"):
        r = bench.tracker_fps(om, cam, dev, counts=(200, 2000, 20000), precision="f64", occlusion=occ)
        print(occ, {n: (round(r[n]["fps"]), round(r[n]["fps_pipelined"]), r[n]["resamplings"], "%.2e" % r[n]["final_position_error_m"]) for n in (200, 2000, 20000)})
