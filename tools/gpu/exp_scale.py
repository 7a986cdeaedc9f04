"""measurement: why is scale_model.solver_only.grid's rank step (in-bench) shorter than the same rank through --emulate-rank-of?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench as B
from sca_amd import solver as S
from sca_amd.distributed import ShardedStepper

scene = B.build_scene(B.WORKLOADS['c4'], 100000)
n = scene['n']
timer = B.Timer(torch, None, 'cuda')
def leg(sol, G, mode, tracked, label):
    cnt = n // G
    sol.set_shard_emulation(True)
    sol.set_shard((G // 2) * cnt, cnt)
    st = ShardedStepper(sol, 0, 1, mode=mode)
    st.begin, st.count = (G // 2) * cnt, cnt
    l = B.timed_leg(sol, scene, st, timer, 100, 20, tracked)
    fl = sol.get_state()['flags']
    print(json.dumps(dict(label=label, G=G, mode=mode, tracked=tracked, ms=round(l['ms_per_step'], 4), agent_steps_per_step=l['agent_steps'] / 100,
                          k1=round(l['k1_ms'], 4), solve=round(l['k_solve_ms'], 4), forms=l['forms'],
                          shard_flags=[int(x) for x in np.bincount(fl[(G // 2) * cnt:(G // 2) * cnt + cnt], minlength=8)[:5]])), flush=True)
sol = B.make_solver(S, scene, 0)
leg(sol, 8, 1, False, 'fresh solver: grid solver-only')
leg(sol, 8, 1, False, 'again')
leg(sol, 8, 0, False, 'kd solver-only')
leg(sol, 8, 1, True, 'grid tracked')
leg(sol, 8, 1, False, 'grid solver-only after a tracked leg')
leg(sol, 2, 1, False, 'G=2 grid solver-only')
sol.set_shard_emulation(False); sol.set_shard(0, n)
st = ShardedStepper(sol, 0, 1, mode=1)
l = B.timed_leg(sol, scene, st, timer, 100, 20, False)
print('full grid solver-only', l['ms_per_step'], l['agent_steps'] / 100)
