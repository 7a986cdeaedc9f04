"""Poses of re-planning agents taken from the benchmark episodes themselves (needs a GPU): c2 (circle, N = 1024) at steps 10 ... 2500,
c5 (take-off / landing, N = 16 384) at steps 5 ... 200, 24 flying SCA agents each -> gpurun_out/episode_poses.json, kept as
tools/data/episode_poses.json.  tools/gen_spec_trees.py --record runs the reference's planner on them (families c2ep / c5ep)."""
import sys, json, numpy as np
sys.path.insert(0, '.')
from sca_amd import scenarios, solver as S
out = {}
rng = np.random.default_rng(3)
for fam, steps in (('c2', [10, 40, 80, 150, 400, 1000, 2500]), ('c5', [5, 20, 60, 120, 200])):
    if fam == 'c2':
        n = 1024; sc = scenarios.circle(n); policy = np.zeros(n, np.uint8)
    else:
        sc = scenarios.takeoff_landing(16384); n = len(sc['start']); policy = np.where(np.arange(n) % 2 == 0, 0, 2).astype(np.uint8)
    sol = S.BatchedSolver(max_agents=n, max_obstacles=max(1, len(sc['obs_radius'])))
    sol.set_obstacles(sc['obs_pos'], sc['obs_radius'])
    sol.set_agents(np.full(n, 0.5), np.ones(n), sc['goal'][:, :3], policy, S.zaxis_flags(sc['start'], sc['goal']), scenarios.max_run_dist(sc['start'], sc['goal']))
    sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
    sol.device_tracker_enable(sc['goal'][:, 3:6], in_pass=True)
    done = 0; cases = []
    for st in steps:
        sol.run_steps(st - done); sol.synchronize(); done = st
        s = sol.get_state()
        ext = np.isin(policy, (0, 5)) & ((s['flags'] & 7) == 0)
        ids = rng.choice(np.nonzero(ext)[0], 24, replace=False)
        for i in ids:
            cases.append(([float(x) for x in s['pos'][i]] + [float(x) for x in s['heading'][i]], [float(x) for x in sc['goal'][i]]))
    out[fam] = cases
json.dump(out, open('gpurun_out/episode_poses.json', 'w'))
print({k: len(v) for k, v in out.items()})
