"""Stress of SCA_NBR_AUTO beyond the test suite (needs a GPU): random scenes -- 40 .. 40 000 agents; uniform boxes from sparse to far
beyond 16 in range, coordinates rounded to 0 / 1 / 2 decimals (equal distances), clusters, flat layers; obstacles; all policies incl. the
Official LP -- stepped side by side on SCA_NBR_KDTREE and SCA_NBR_AUTO with random burst lengths (1: every pass builds its own tree;
more: the next pass's tree is built ahead behind the integrate stage).  States, action rows, lists entry for entry and the carried
permutation must be equal after every burst.    python tools/fuzz_auto.py <seed> <scenes>"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from sca_amd import scenarios, solver as S

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
nscenes = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
agent_steps = 0
t0 = time.time()
for sc_i in range(nscenes):
    n = int(rng.choice([40, 200, 1000, 3000, 9000, 20000, 40000]))
    dens = float(rng.choice([0.3, 1.0, 3.0, 8.0, 30.0]))              # expected agents within neighborDist
    side = (n * 4189.0 / dens) ** (1 / 3) / 2
    start = np.zeros((n, 6)); goal = np.zeros((n, 6))
    start[:, :3] = rng.uniform(-side, side, (n, 3)); goal[:, :3] = rng.uniform(-side, side, (n, 3))
    kind = rng.integers(0, 5)
    if kind == 1: start[:, :3] = np.round(start[:, :3], int(rng.integers(0, 3)))            # equal distances
    if kind == 2: start[:, 2] = np.round(start[:, 2] / 5.0) * 5.0                              # flat layers
    if kind == 3: start[: n // 2, :3] = start[n // 2: 2 * (n // 2), :3] * 0.05               # a cluster in the middle
    start[:, 2] += side + 5.0; goal[:, 2] += side + 5.0
    start[:, 3] = rng.uniform(0, 2 * np.pi, n)
    pol = rng.choice([1, 2, 3, 4], n).astype(np.uint8) if rng.random() < 0.7 else np.full(n, int(rng.choice([1, 2, 3, 4])), np.uint8)
    m = int(rng.choice([0, 0, 5, 40]))
    obs_pos = rng.uniform(-side, side, (m, 3)) + np.array([0, 0, side + 5.0]); obs_r = rng.uniform(0.2, 2.0, m)
    zaxis = S.zaxis_flags(start, goal); mrd = scenarios.max_run_dist(start, goal)
    def mk():
        sol = S.BatchedSolver(max_agents=n, max_obstacles=max(1, m))
        sol.set_obstacles(obs_pos, obs_r)
        sol.set_agents(np.full(n, 0.5), np.ones(n), goal[:, :3], pol, zaxis, mrd)
        sol.set_state(start[:, :3], np.zeros((n, 3), np.float32), start[:, 3:6], np.zeros(n, np.uint8))
        return sol
    a, b = mk(), mk()
    ok = True
    steps = 0
    while steps < 24 and ok:
        burst = int(rng.integers(1, 8))
        a.run_steps(burst, S.NBR_KDTREE); b.run_steps(burst, S.NBR_AUTO)
        a.synchronize(); b.synchronize()
        steps += burst
        sa, sb = a.get_state(), b.get_state()
        na, nb = a.neighbors(), b.neighbors()
        same = all(np.array_equal(sa[k], sb[k]) for k in ('pos', 'vel', 'heading', 'flags', 'total_dist')) and \
            all(np.array_equal(na[k], nb[k]) for k in ('nbr_valid', 'nbr_n', 'nbr_id', 'nbr_kind', 'nbr_dsq')) and \
            np.array_equal(a.actions(), b.actions()) and np.array_equal(a.get_kd_perm(), b.get_kd_perm())
        if not same:
            ok = False
            print('MISMATCH scene', sc_i, 'n', n, 'density', dens, 'kind', int(kind), 'after', steps, 'steps', flush=True)
    bad += not ok
    agent_steps += n * steps
    a.close(); b.close()
print('scenes', nscenes, 'bad', bad, 'agent-steps', agent_steps, 'seconds %.0f' % (time.time() - t0))
