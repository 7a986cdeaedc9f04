"""measurement: distribution of the planner's search lengths (candidate radii tried) over the c4 episode's early steps"""
import ctypes as C
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench as B
from sca_amd import _lib, solver as S

w = sys.argv[1] if len(sys.argv) > 1 else 'c4'
scene = B.build_scene(B.WORKLOADS[w], B.WORKLOADS[w]['n'])
sol = B.make_solver(S, scene, 0)
B.reset_state(sol, scene)
sol.device_tracker_enable(scene['sc']['goal'][:, 3:6])
n = scene['n']
sel = np.arange(0, n, max(1, n // 2000))
o = np.zeros(24)
rows = []
done = 0
for upto in (1, 2, 3, 5, 8, 12, 16, 20, 25, 30, 40, 60, 100, 200):
    sol.run_steps(upto - done)
    sol.synchronize()
    done = upto
    it = []
    for i in sel:
        sol.L.sca_device_tracker_debug(sol.ctx, int(i), _lib.ptr(o, C.c_double))
        it.append(o[22] / 64)
    it = np.array(it)
    rows.append(dict(step=upto, mean=float(it.mean()), p50=float(np.percentile(it, 50)), p90=float(np.percentile(it, 90)), p95=float(np.percentile(it, 95)),
                     p99=float(np.percentile(it, 99)), max=float(it.max()), buckets=np.bincount(np.clip((it.astype(int) - 40) >> 3, 0, 11), minlength=12).tolist()))
    print(json.dumps(rows[-1]), flush=True)
