"""Test-only backend for sca_amd.distributed.ShardedStepper: the same step_begin / exchange / step_end protocol as
libsca_hip, with the CPU oracle doing the per-agent work.  Lets the 2-rank gloo test prove that exchanging the moved
48-byte public records once per step is sufficient for bit-identical results."""
import numpy as np
import torch

from oracle import oracle as orc
from sca_amd import scenarios, solver as S

REC = np.dtype([('pos', '<f8', 3), ('vel', '<f4', 3), ('flags', '<u4'), ('radius', '<f8')])
assert REC.itemsize == 48


def scene(n=96):
    sc = scenarios.random_cube(n, seed=21)
    rng = np.random.default_rng(4)
    d = sc['goal'][:, :3] - sc['start'][:, :3]
    v = d / np.linalg.norm(d, axis=1, keepdims=True) + rng.normal(0, 0.2, (n, 3))
    return dict(n=n, pos=sc['start'][:, :3].copy(), vel=(0.8 * v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32),
                heading=sc['start'][:, 3:6].copy(), goal=sc['goal'][:, :3].copy(), radius=np.full(n, 0.5),
                pref_speed=np.full(n, 1.0), policy=np.full(n, 3, np.uint8), zaxis=np.zeros(n, np.uint8),
                max_run_dist=scenarios.max_run_dist(sc['start'], sc['goal']))


def reference_run(sc, steps):
    n = sc['n']
    pos, vel, head = sc['pos'].copy(), sc['vel'].copy(), sc['heading'].copy()
    flags = np.zeros(n, np.uint8); td = np.zeros(n); sn = np.zeros(n, np.int32); perm = np.arange(n, dtype=np.int32)
    e3, e0 = np.zeros((0, 3)), np.zeros(0)
    for _ in range(steps):
        r = orc.policy_step(pos, vel, head, sc['radius'], sc['pref_speed'], flags, sc['goal'], sc['policy'], sc['zaxis'],
                            np.zeros((n, 3)), np.zeros(n, np.uint8), perm, e3, e0)
        perm = r['perm']
        u = orc.env_update(pos, vel, head, sc['radius'], r['flags'], sc['goal'], r['action'], td, sc['max_run_dist'], sn, e3, e0)
        pos, vel, head, flags, td, sn = u['pos'], u['vel'], u['heading'], u['flags'], u['total_dist'], u['step_num']
    return dict(pos=pos, vel=vel, flags=flags)


class CheckerBackend:
    def __init__(self, sc):
        self.sc = sc
        self.n = n = sc['n']
        self.pos, self.vel, self.heading = sc['pos'].copy(), sc['vel'].copy(), sc['heading'].copy()
        self.flags = np.zeros(n, np.uint8)
        self.td = np.zeros(n); self.sn = np.zeros(n, np.int32); self.perm = np.arange(n, dtype=np.int32)
        self.begin, self.count = 0, n
        self.moved = np.zeros(n, REC)
        self._t = torch.from_numpy(self.moved.view(np.uint8).reshape(-1))

    def set_shard(self, begin, count):
        self.begin, self.count = begin, count

    def moved_records(self):
        per = self.count * REC.itemsize
        return self._t, self._t[self.begin * REC.itemsize: self.begin * REC.itemsize + per]

    def step_begin(self, mode=0):
        sc, n = self.sc, self.n
        e3, e0 = np.zeros((0, 3)), np.zeros(0)
        r = orc.policy_step(self.pos, self.vel, self.heading, sc['radius'], sc['pref_speed'], self.flags, sc['goal'],
                            sc['policy'], sc['zaxis'], np.zeros((n, 3)), np.zeros(n, np.uint8), self.perm, e3, e0)
        self.perm = r['perm']
        lo, hi = self.begin, self.begin + self.count
        self.old_pos = self.pos.copy()
        self.flags[lo:hi] = r['flags'][lo:hi]
        # integrate only the shard (update_velocitie, mampenv.py:83-105) through the oracle on a copy
        far = np.full(n, 1e9)
        u = orc.env_update(self.pos, self.vel, self.heading, np.zeros(n), self.flags | 1, sc['goal'] + 1e6, r['action'],
                           self.td, far, self.sn, e3, e0)            # radius 0 / goal far: pure kinematics, no flags
        self.moved['pos'][lo:hi] = u['pos'][lo:hi]
        self.moved['vel'][lo:hi] = u['vel'][lo:hi]
        self.moved['flags'][lo:hi] = self.flags[lo:hi]
        self.moved['radius'][lo:hi] = sc['radius'][lo:hi]
        self.heading[lo:hi] = u['heading'][lo:hi]
        self.td[lo:hi] = u['total_dist'][lo:hi]
        self.sn[lo:hi] += (self.flags[lo:hi] & 1) == 0

    def step_end(self):
        sc, n = self.sc, self.n
        new_pos = self.moved['pos'].copy()
        lo, hi = self.begin, self.begin + self.count

        def l3(a, b):
            return np.array([round(float(x), 5) for x in np.sqrt(((a - b) ** 2).sum(-1))])
        flags = self.moved['flags'].astype(np.uint8)
        for a in range(lo, hi):                                       # check_agent_state, mampenv.py:68-79
            if flags[a] & 1:
                continue
            rs = sc['radius'][a] + sc['radius']
            c = l3(new_pos[a], new_pos) <= rs
            later = np.arange(n) > a
            c |= later & (l3(new_pos[a], self.old_pos) <= rs)
            c |= (~later) & (l3(new_pos, self.old_pos[a]) <= rs)
            c[a] = False
            if c.any():
                flags[a] |= 2
            if self.td[a] > sc['max_run_dist'][a]:
                flags[a] |= 4
        goal_hit = l3(new_pos, sc['goal']) <= 0.5                     # is_done, replicated for every agent
        flags[goal_hit] |= 1
        keep = np.ones(n, bool); keep[lo:hi] = False
        flags[keep] = (self.moved['flags'][keep].astype(np.uint8)) | goal_hit[keep]
        self.pos = new_pos
        self.vel = self.moved['vel'].copy()
        self.flags = flags

    def run_steps(self, steps, mode=0):
        for _ in range(steps):
            self.step_begin(mode)
            self.step_end()

    def synchronize(self):
        pass
