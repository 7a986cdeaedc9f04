"""One-off check beyond the suite's sizes (needs a GPU; drives the oracle, hence under tests/): swarms of 1 and 3 million agents (the kd level
passes take their chunks by ticket beyond ~2 million) -- two steps: the kd-mode policy pass against the oracle's (action rows, kd permutation,
neighbour lists EQUAL), the resident SCA_NBR_AUTO step against the kd context's state.    python tests/big_swarm_check.py [n ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc                      # noqa: E402
from sca_amd import solver as S                       # noqa: E402

orc.build()
threads = min(16, os.cpu_count() or 1)
for n in [int(a) for a in sys.argv[1:]] or [1_000_000, 3_000_000]:
    rng = np.random.default_rng(n)
    side = 0.5 * (n / 0.004) ** (1.0 / 3.0)                                # ~17 agents within neighborDist of one another
    pos = rng.uniform(-side, side, (n, 3)); pos[:, 2] += side + 5.0
    goal = rng.uniform(-side, side, (n, 3)); goal[:, 2] += side + 5.0
    head = np.zeros((n, 3)); head[:, 0] = rng.uniform(0, 2 * np.pi, n)
    v = rng.normal(0, 1, (n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True); v *= rng.uniform(0.2, 1.0, (n, 1))
    policy = rng.choice([1, 2, 3, 4], n).astype(np.uint8)
    flags = np.zeros(n, np.uint8)
    rad = np.full(n, 0.5); ps = np.ones(n); z = np.zeros(n, np.uint8); mrd = np.full(n, 1e9)
    e3, e0 = np.zeros((0, 3)), np.zeros(0)

    def mk():
        sol = S.BatchedSolver(max_agents=n, max_obstacles=1)
        sol.set_obstacles(e3, e0)
        sol.set_agents(rad, ps, goal, policy, z, mrd)
        sol.set_state(pos, v.astype(np.float32), head, flags, np.zeros(n), np.zeros(n, np.int32))
        return sol

    a, b = mk(), mk()
    perm = np.arange(n, dtype=np.int32)
    bad = []
    for t in range(2):
        st = a.get_state()
        # one policy pass in kd mode against the oracle's (the oracle's env update is the reference's all-pairs loop: not at this size)
        t0 = time.time()
        r = orc.policy_step(st['pos'], st['vel'], st['heading'], rad, ps, st['flags'], goal, policy, z, np.zeros((n, 3)), np.zeros(n, np.uint8), perm, e3, e0,
                            nthreads=threads)
        t_or = time.time() - t0
        perm = r['perm']
        t0 = time.time(); a.policy_pass(S.NBR_KDTREE); a.synchronize(); t_kd = time.time() - t0
        nb = a.neighbors()
        for k, got, want in (('action', a.actions(), r['action']), ('perm', a.get_kd_perm(), perm), ('nbr_n', nb['nbr_n'], r['nbr_n']), ('nbr_id', nb['nbr_id'], r['nbr_id']),
                             ('nbr_valid', nb['nbr_valid'], r['nbr_valid'])):
            if not np.array_equal(got, want):
                bad.append(('kd vs oracle', t, k, int((np.asarray(got) != np.asarray(want)).reshape(n, -1).any(axis=1).sum())))
        a.env_update(False)
        # the same step resident in SCA_NBR_AUTO: the state after it equals the kd context's
        t0 = time.time(); b.run_steps(1, S.NBR_AUTO); b.synchronize(); t_auto = time.time() - t0
        ga, gb = a.get_state(), b.get_state()
        for k in ('pos', 'vel', 'heading', 'flags', 'total_dist', 'step_num'):
            if not np.array_equal(ga[k], gb[k]):
                bad.append(('auto vs kd', t, k, int((ga[k] != gb[k]).reshape(n, -1).any(axis=1).sum())))
        if not np.array_equal(a.get_kd_perm(), b.get_kd_perm()):
            bad.append(('auto vs kd', t, 'perm'))
        print(f'n {n} step {t}: kd pass {t_kd * 1e3:.2f} ms  auto step {t_auto * 1e3:.2f} ms  oracle pass ({threads} threads) {t_or:.1f} s  '
              f'lists longer than 16 cut: {int((r["nbr_n"] == 16).sum())}  collided {int((ga["flags"] & 2).astype(bool).sum())}', flush=True)
    print('n', n, 'BAD' if bad else 'equal', bad[:8], flush=True)
    a.close(); b.close()
