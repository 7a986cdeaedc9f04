"""measurement: wall time of each of the first steps of a workload (one resident step per call + synchronize), and the forms each pass ran"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from sca_amd import solver as S
wname = sys.argv[1] if len(sys.argv) > 1 else 'c4'
w = B.WORKLOADS[wname]
scene = B.build_scene(w, w['n'])
sol = B.make_solver(S, scene, 0)
rows = []
for rep in range(2):
    B.reset_state(sol, scene)
    if w['policy'] in ('sca', 'mixed'):
        sol.device_tracker_enable(scene['sc']['goal'][:, 3:6])
    sol.synchronize()
    ts = []
    for t in range(40):
        t0 = time.perf_counter()
        sol.run_steps(1, 0)
        sol.synchronize()
        ts.append(round((time.perf_counter() - t0) * 1e3, 3))
    rows.append(ts)
    print(json.dumps({'rep': rep, 'ms': ts}))
    try:
        print('replans', sol.device_tracker_replans()[:1] if hasattr(sol, 'device_tracker_replans') else None)
    except Exception as e:
        print('n/a', e)
