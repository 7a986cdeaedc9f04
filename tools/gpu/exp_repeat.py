"""measurement: the driver's leg (5 warm-up + 20 timed steps from the start state) repeated in one process -- is the slower first
leg the episode's phase (every repeat equally slow) or the GPU's clocks / caches (only the first one slow)?"""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
from sca_amd import solver as S
from sca_amd.distributed import ShardedStepper

w = sys.argv[1] if len(sys.argv) > 1 else 'c4'
scene = B.build_scene(B.WORKLOADS[w], B.WORKLOADS[w]['n'])
sol = B.make_solver(S, scene, 0)
timer = B.Timer(torch, None, 'cuda')
st = ShardedStepper(sol, 0, 1, mode=0)
tracked = B.WORKLOADS[w]['policy'] in ('sca', 'mixed')
rows = []
for rep in range(4):
    leg = B.timed_leg(sol, scene, st, timer, 20, 5, tracked)
    rows.append(dict(rep=rep, ms_per_step=leg['ms_per_step'], replan_ms=leg['replan_ms'], plans=leg['plans']))
leg = B.timed_leg(sol, scene, st, timer, 20, 40, tracked)
rows.append(dict(rep='w40', ms_per_step=leg['ms_per_step'], replan_ms=leg['replan_ms'], plans=leg['plans']))
print(json.dumps(rows))
