"""Stress beyond the test suite (needs a GPU; test infrastructure: it drives the oracle, hence it lives under tests/): random scenes at the
sizes where the library switches kernel forms -- 1 .. 40 000 agents across k_solve_fb (<= 2048), k_kd_top (<= 4096), k_neighbors_kd / kd4
(6144), k_action_fb / k_lp (16 384) -- all six policies mixed or one policy for everybody, obstacles, agents done from the start, dense and
sparse boxes, SCA_NBR_KDTREE and SCA_NBR_AUTO, FREE-RUNNING from the scene's state: after every resident step flags, step counts, the kd
permutation, float32 velocities, positions, headings and travelled distance must EQUAL the oracle's (tests/test_gpu_parity.py's fuzz family,
whose scenes stop at 1600 agents and 6 steps).

    python tests/fuzz_oracle.py <seed> <scenes> [max agents per cubic metre: 0.2 (default; the oracle needs minutes for the largest dense scenes), 0.02]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc                      # noqa: E402
from sca_amd import solver as S                       # noqa: E402

orc.build()
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nscenes = int(sys.argv[2]) if len(sys.argv) > 2 else 40
max_density = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2
rng = np.random.default_rng(seed0)
threads = min(16, os.cpu_count() or 1)
bad = 0
agent_steps = 0
t0 = time.time()
for sc_i in range(nscenes):
    n = int(rng.choice([1, 7, 100, 1000, 2047, 2049, 4096, 4097, 6143, 6145, 10000, 16383, 16385, 25000, 40000]))
    m = int(rng.choice([0, 0, 3, 40, 400]))
    density = float(rng.choice([dd for dd in (0.002, 0.02, 0.2) if dd <= max_density]))                        # agents per cubic metre: <1, ~8, ~80 in range of one another
    side = max(2.0, 0.5 * (n / density) ** (1.0 / 3.0))
    steps = int(rng.integers(4, 13)) if n <= 10000 else int(rng.integers(3, 7))
    mode = S.NBR_AUTO if rng.random() < 0.5 else S.NBR_KDTREE
    pos = rng.uniform(-side, side, (n, 3))
    pos[:, 2] = np.abs(pos[:, 2]) + float(rng.choice([0.0, 1.0, 20.0]))
    goal = rng.uniform(-side, side, (n, 3))
    goal[:, 2] = np.abs(goal[:, 2]) + 1.0
    if rng.random() < 0.3:
        goal[: n // 2, :2] = pos[: n // 2, :2]                             # is_zAxis agents (scaPolicy.py:188-190)
    head = np.zeros((n, 3))
    head[:, 0] = rng.uniform(0, 2 * np.pi, n)
    head[:, 1] = rng.uniform(-0.5, 0.5, n)
    v = rng.normal(0, 1, (n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v *= rng.uniform(0, 1, (n, 1))
    if rng.random() < 0.3:
        v[rng.random(n) < 0.3] = 0.0                                       # bootstrap branch for some
    one = int(rng.integers(0, 7))
    policy = (rng.integers(0, 6, n) if one == 6 else np.full(n, one)).astype(np.uint8)
    flags = ((rng.random(n) < 0.05) * rng.choice([1, 2, 4], n)).astype(np.uint8)
    obs_pos = rng.uniform(-side, side, (m, 3))
    obs_pos[:, 2] = np.abs(obs_pos[:, 2])
    obs_radius = rng.choice([0.2, 1.0, 2.0], m)
    radius = rng.choice([0.3, 0.5, 1.0], n)
    pref_speed = rng.choice([1.0, 1.0, 0.8, 1.5], n)
    vpref = np.trunc(rng.normal(0, 0.6, (n, 3)) * 1e5) / 1e5               # "tracker output" for SCA / RVO3D+Dubins
    vmode = np.isin(policy, (0, 5)).astype(np.uint8)
    mrd = 3.0 * np.linalg.norm(pos - goal, axis=1) + 1.0
    zaxis = S.zaxis_flags(np.concatenate([pos, head], 1), np.concatenate([goal, np.zeros((n, 3))], 1))
    sol = S.BatchedSolver(max_agents=n, max_obstacles=max(1, m))
    sol.set_obstacles(obs_pos, obs_radius)
    sol.set_agents(radius, pref_speed, goal, policy, zaxis, mrd)
    sol.set_vpref(vpref, vmode)
    p, ve, he, fl = pos.copy(), v.astype(np.float32), head.copy(), flags.copy()
    td = np.zeros(n)
    sn = np.zeros(n, np.int32)
    perm = np.arange(n, dtype=np.int32)
    sol.set_state(p, ve, he, fl, td, sn)
    sol.set_kd_perm(perm)
    ok = True
    for t in range(steps):
        sol.run_steps(1, mode)
        sol.synchronize()
        g = sol.get_state()
        r = orc.policy_step(p, ve, he, radius, pref_speed, fl, goal, policy, zaxis, vpref, vmode, perm, obs_pos, obs_radius, nthreads=threads)
        perm = r['perm']
        u = orc.env_update(p, ve, he, radius, r['flags'], goal, r['action'], td, mrd, sn, obs_pos, obs_radius)
        p, ve, he, fl, td, sn = u['pos'], u['vel'], u['heading'], u['flags'], u['total_dist'], u['step_num']
        agent_steps += n
        what = [k for k, a, b in (('flags', g['flags'], fl), ('step_num', g['step_num'], sn), ('perm', sol.get_kd_perm(), perm), ('vel', g['vel'], ve),
                                  ('pos', g['pos'], p), ('heading', g['heading'], he), ('total_dist', g['total_dist'], td)) if not np.array_equal(a, b)]
        if what:
            ok = False
            print('MISMATCH scene', sc_i, 'n', n, 'm', m, 'density', density, 'policy', one, 'mode', mode, 'step', t, what,
                  int((g['pos'] != p).any(axis=1).sum()), 'positions differ')
            break
    st = sol.diag()['status']
    if ok and mode == S.NBR_KDTREE and int((st & (16 | 64 | 128)).any()):
        ok = False
        print('STATUS BITS scene', sc_i, np.unique(st))
    bad += not ok
    sol.close()
    print('scene', sc_i, 'n', n, 'm', m, 'density', density, 'policy', one, 'mode', mode, 'steps', steps, 'ok' if ok else 'BAD', '%.0f s' % (time.time() - t0), flush=True)
print('scenes', nscenes, 'bad', bad, 'agent-steps', agent_steps, 'seconds %.0f' % (time.time() - t0))
