#!/usr/bin/env python3
"""how many agents does the AUTO grid query hand to the kd query at c4 (ring density)?  SCA_AUTO_BACKOFF_DIV=1 keeps AUTO from backing off."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ['SCA_AUTO_BACKOFF_DIV'] = '1'
import bench as B  # noqa: E402
from sca_amd import solver as S  # noqa: E402

out = {}
for wname in ('c4', 'c2', 'c3', 'c5'):
    w = B.WORKLOADS[wname]
    scene = B.build_scene(w, w['n'])
    sol = B.make_solver(S, scene, 0)
    B.reset_state(sol, scene)
    sol.run_steps(10, S.NBR_AUTO)
    sol.synchronize()
    sol.auto_stats(reset=True)
    sol.run_steps(20, S.NBR_AUTO)
    sol.synchronize()
    st = sol.auto_stats()
    st['agents'] = scene['n']
    st['listed_frac_of_swarm'] = st['listed_per_pass_mean'] / scene['n']
    out[wname] = st
    sol.close()
print(json.dumps(out, indent=1))
