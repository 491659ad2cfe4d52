import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import peer_trail_worker as w
if __name__ == "__main__":
    for shared in (False, True):
        got = w.run(27500 + 10 * int(shared), 0, "device", shared)
        for rk, res, planes, area, state in got:
            print("shared", shared, "rank", rk, "state", state, "area", area, "nan per step", [int(np.isnan(ll).sum()) for ll, _ in res])
