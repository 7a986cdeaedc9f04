"""measurement: is a resident burst of short steps bound by the HOST's enqueue rate?  time until sca_run_steps returns (everything enqueued)
against the time until the device is through"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from sca_amd import solver as S
for wname, nbr in (('c3', 'auto'), ('c3', 'kd'), ('c3', 'grid'), ('c3lp', 'auto'), ('c2', 'kd'), ('c5', 'kd'), ('c4', 'kd')):
    w = B.WORKLOADS[wname]
    scene = B.build_scene(w, w['n'])
    sol = B.make_solver(S, scene, 0)
    B.reset_state(sol, scene)
    tracked = w['policy'] in ('sca', 'mixed')
    if tracked:
        sol.device_tracker_enable(scene['sc']['goal'][:, 3:6])
    mode = B.NBR[nbr]
    sol.run_steps(30, mode); sol.synchronize()
    k = 200 if wname != 'c4' else 50
    t0 = time.perf_counter()
    sol.run_steps(k, mode)
    t1 = time.perf_counter()
    sol.synchronize()
    t2 = time.perf_counter()
    print(json.dumps(dict(workload=wname, nbr=nbr, steps=k, host_enqueue_us_per_step=round((t1 - t0) / k * 1e6, 2), total_us_per_step=round((t2 - t0) / k * 1e6, 2),
                          device_still_busy_after_enqueue_us=round((t2 - t1) * 1e6, 1))), flush=True)
    sol.close()
