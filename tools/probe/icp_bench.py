#!/usr/bin/env python3
"""rcv_icp_point_to_point: hashed-grid neighbour search against the exhaustive one (RCV_ICP_NO_GRID=1) on a LINEMOD-sized case."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_registration as TR
from rcvpose_amd.registration import ICPConvergenceCriteria, registration_icp
rng = np.random.default_rng(0)
model = TR._model(rng, 6000)
Rg, tg = TR._rot(rng, 40), np.array([20.0, -30.0, 900.0])
scene_full = TR._model(rng, 30000)
vis = (scene_full @ Rg.T)[:, 2] < 10
scene = scene_full[vis] @ Rg.T + tg + rng.normal(size=(vis.sum(), 3)) * 0.3
init = np.eye(4); init[:3, :3] = TR._rot(rng, 2.0) @ Rg; init[:3, 3] = tg + rng.normal(size=3) * 2.0
ms, sc = torch.from_numpy(model).cuda(), torch.from_numpy(scene).cuda()
print("model", model.shape[0], "scene", scene.shape[0])
for reach in (3.0, 6.0, 12.0, 40.0):
    for env in ("0", "1"):
        os.environ["RCV_ICP_NO_GRID"] = env
        r = registration_icp(ms, sc, reach, init, ICPConvergenceCriteria())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): r = registration_icp(ms, sc, reach, init, ICPConvergenceCriteria())
        torch.cuda.synchronize()
        print("reach %5.1f mm  %s  %.2f ms per call  (%d iterations, fitness %.3f)" % (reach, "exhaustive" if env == "1" else "grid      ", (time.perf_counter() - t0) * 100, r.iterations, r.fitness))
