#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container, never on the GPU box).

Imports the read-only reference checkout (default /root/reference) and records, for a set of
scenarios, the exact inputs and outputs of the per-agent hot path
(neighbour selection -> cone / half-space construction -> velocity selection) plus the env update,
as small ``.npz`` fixtures under ``tests/golden/``.  Only *data* is written (inputs / expected
outputs); no reference source travels.

Reference entry points that are wrapped (module-level monkeypatching, the reference is not edited):
  mamp/envs/mampenv.py:22 MACAEnv.step                -> per-step pre/post state
  mamp/policies/*:find_next_action                    -> action[7] (via all_actions rows)
  mamp/policies/*:compute_v_pref                      -> v_pref input of the solver
  mamp/policies/*:computeNeighbors                    -> neighbour list (ids, kinds, distSq, order)
  mamp/policies/*:intersect / compute_newV_is_suit    -> suitable count, fallback flag
  orca3dPolicyOfficial.linearProgram3 / 4             -> planeFail, LP4-ran flag

Usage:  python tools/gen_golden.py [--only NAME ...] [--ref /root/reference]
"""
import argparse
import contextlib
import io
import math
import os
import random
import sys
import time

import numpy as np

POL_SCA, POL_RVO, POL_SRVO, POL_ORCA, POL_ORCA_LP, POL_RVO_DUBINS = 0, 1, 2, 3, 4, 5
K = 16


def _import_reference(ref):
    sys.path.insert(0, ref)
    sys.path.insert(0, os.path.join(ref, 'run_example'))
    import matplotlib
    matplotlib.use('Agg')
    import mamp.agents.agent as agent_mod
    # logging-only shim: Agent.to_vector appends a pandas row per agent per step (agent.py:126-147),
    # DataFrame.append no longer exists in pandas 2 -> make the logger a no-op (off the hot path).
    agent_mod.Agent.to_vector = lambda self: None
    import mamp.envs.mampenv as env_mod
    from mamp.policies.sca import scaPolicy, rvo3dDubinsPolicy
    from mamp.policies import rvo3dPolicy, srvo3dPolicy, orca3dPolicy, orca3dPolicyOfficial
    mods = {POL_SCA: scaPolicy, POL_RVO: rvo3dPolicy, POL_SRVO: srvo3dPolicy, POL_ORCA: orca3dPolicy,
            POL_ORCA_LP: orca3dPolicyOfficial, POL_RVO_DUBINS: rvo3dDubinsPolicy}
    classes = {POL_SCA: scaPolicy.SCAPolicy, POL_RVO: rvo3dPolicy.RVO3DPolicy, POL_SRVO: srvo3dPolicy.SRVO3DPolicy,
               POL_ORCA: orca3dPolicy.ORCA3DPolicy, POL_ORCA_LP: orca3dPolicyOfficial.ORCA3DPolicy,
               POL_RVO_DUBINS: rvo3dDubinsPolicy.RVO3dDubinsPolicy}
    return agent_mod, env_mod, mods, classes


class Recorder:
    """Per-step scratch filled by the wrappers."""

    def __init__(self, n):
        self.n = n
        self.reset()

    def reset(self):
        n = self.n
        self.vpref = np.full((n, 3), np.nan)
        self.called = np.zeros(n, np.uint8)
        self.nbr_valid = np.zeros(n, np.uint8)
        self.nbr_n = np.zeros(n, np.int32)
        self.nbr_id = np.full((n, K), -1, np.int32)
        self.nbr_kind = np.zeros((n, K), np.uint8)
        self.nbr_dsq = np.zeros((n, K))
        self.n_suit = np.full(n, -1, np.int32)
        self.fallback = np.zeros(n, np.uint8)
        self.plane_fail = np.full(n, -1, np.int32)
        self.lp4 = np.zeros(n, np.uint8)
        self.vpost = np.full((n, 3), np.nan)


REC = None
_SUIT_COUNT = [0]


def _install_wrappers(mods):
    for pid, mod in mods.items():
        # compute_v_pref: signature (agent) for the Dubins policies, (goal, agent) otherwise.
        orig_vp = mod.compute_v_pref

        def vp(*args, _o=orig_vp):
            out = _o(*args)
            ag = args[-1]
            REC.vpref[ag.id] = out
            REC.called[ag.id] = 1
            return out
        mod.compute_v_pref = vp

        orig_cn = mod.computeNeighbors

        def cn(agent, kdTree, _o=orig_cn):
            _o(agent, kdTree)
            i = agent.id
            REC.nbr_valid[i] = 1
            REC.nbr_n[i] = len(agent.neighbors)
            for k, (obj, dsq) in enumerate(agent.neighbors):
                REC.nbr_id[i, k] = obj.id
                REC.nbr_kind[i, k] = 1 if obj.is_obstacle else 0
                REC.nbr_dsq[i, k] = dsq
        mod.computeNeighbors = cn

        if hasattr(mod, 'intersect'):
            orig_suit = mod.compute_newV_is_suit

            def suit(*args, _o=orig_suit):
                r = _o(*args)
                if r:
                    _SUIT_COUNT[0] += 1
                return r
            mod.compute_newV_is_suit = suit
            orig_int = mod.intersect
            if pid == POL_ORCA:
                def inter(agent, v_pref, planes, _o=orig_int):
                    _SUIT_COUNT[0] = 0
                    out = _o(agent, v_pref, planes)
                    REC.n_suit[agent.id] = _SUIT_COUNT[0]
                    REC.fallback[agent.id] = 1 if _SUIT_COUNT[0] == 0 else 0
                    REC.vpost[agent.id] = out
                    return out
            else:
                def inter(v_pref, rvo, agent, _o=orig_int):
                    _SUIT_COUNT[0] = 0
                    out = _o(v_pref, rvo, agent)
                    REC.n_suit[agent.id] = _SUIT_COUNT[0]
                    REC.fallback[agent.id] = 1 if _SUIT_COUNT[0] == 0 else 0
                    REC.vpost[agent.id] = out
                    return out
            mod.intersect = inter
    # LP bookkeeping for the "Official" ORCA policy
    cls = mods[POL_ORCA_LP].ORCA3DPolicy
    orig_fna = cls.find_next_action
    orig_lp3 = cls.linearProgram3
    orig_lp4 = cls.linearProgram4
    cur = [None]

    def fna(self, dict_comm, agent, kdTree):
        cur[0] = agent.id
        self._top = True
        return orig_fna(self, dict_comm, agent, kdTree)

    def lp3(self, planes, maxSpeed, vel_pref, dir_opt=False):
        r = orig_lp3(self, planes, maxSpeed, vel_pref, dir_opt)
        if not dir_opt:
            REC.plane_fail[cur[0]] = r
        return r

    def lp4(self, planes, beginPlane, radius):
        REC.lp4[cur[0]] = 1
        return orig_lp4(self, planes, beginPlane, radius)
    cls.find_next_action = fna
    cls.linearProgram3 = lp3
    cls.linearProgram4 = lp4


def _flags(a):
    return (1 if a.is_at_goal else 0) | (2 if a.is_collision else 0) | (4 if a.is_out_of_max_time else 0)


def _snapshot(agents):
    n = len(agents)
    pos = np.array([a.pos_global_frame for a in agents], dtype=np.float64).reshape(n, 3)
    vel = np.array([np.asarray(a.vel_global_frame, dtype=np.float32) for a in agents], dtype=np.float32).reshape(n, 3)
    head = np.array([np.asarray(a.heading_global_frame, dtype=np.float64) for a in agents]).reshape(n, 3)
    fl = np.array([_flags(a) for a in agents], np.uint8)
    td = np.array([a.total_dist for a in agents], np.float64)
    goal = np.array([a.goal_global_frame for a in agents], dtype=np.float64).reshape(n, 3)
    return pos, vel, head, fl, td, goal


ATTR_KEYS = ('maxNeighbors', 'neighborDist', 'timeStep', 'timeHorizon', 'maxSpeed', 'min_heading_change', 'max_heading_change',
             'turning_radius', 'dt_nominal')


def _apply_attrs(agents, attrs):
    """F16: solver attributes changed after Agent.__init__ (agent.py:24-41), the way a user of the reference does it.
    Values are scalars or per-agent sequences; 'pitchlims' is a (lo, hi) pair and changes ONLY agent.pitchlims (what the Dubins planner
    reads, scaPolicy.py:95) -- agent.max_heading_change (util.py:17) was derived from it inside __init__ and is its own key here."""
    n = len(agents)
    if callable(attrs):                                  # F17: per-agent values drawn once the scene knows its agent count
        attrs = attrs(n)
    for name, val in (attrs or {}).items():
        if name == 'pitchlims':
            v = np.asarray(val, dtype=np.float64)
            for i, a in enumerate(agents):                   # one pair for everybody, or one pair per agent ([n, 2])
                a.pitchlims = [float(v[i, 0]), float(v[i, 1])] if v.ndim == 2 else [float(v[0]), float(v[1])]
            continue
        vals = np.broadcast_to(np.asarray(val), (n,))
        for a, v in zip(agents, vals):
            setattr(a, name, int(v) if name == 'maxNeighbors' else float(v))


def _attr_arrays(agents):
    out = {f'attr_{k}': np.array([getattr(a, k) for a in agents], np.int32 if k == 'maxNeighbors' else np.float64) for k in ATTR_KEYS}
    out['attr_pitch_lo'] = np.array([a.pitchlims[0] for a in agents], np.float64)
    out['attr_pitch_hi'] = np.array([a.pitchlims[1] for a in agents], np.float64)
    return out


def run_env_episode(agent_mod, env_mod, classes, name, pos, goal, policy_ids, obstacles_spec, max_steps,
                    record_every=1, record_first=0, radius=0.5, pref_speed=1.0, outdir='tests/golden', attrs=None):
    """radius / pref_speed may be scalars or per-agent sequences; attrs: see _apply_attrs (recorded as attr_* arrays)."""
    """Runs MACAEnv.step (mampenv.py:22) and records every `record_every`-th step (and the first
    `record_first` steps) completely."""
    global REC
    from mamp.agents.obstacle import Obstacle
    n = len(pos)
    radius_l = [float(x) for x in np.broadcast_to(radius, (n,))]
    ps_l = [float(x) for x in np.broadcast_to(pref_speed, (n,))]
    agents = [agent_mod.Agent(start_pos=list(pos[i]), goal_pos=list(goal[i]), vel=[0.0, 0.0, 0.0], radius=radius_l[i],
                              pref_speed=ps_l[i], policy=classes[int(policy_ids[i])], id=i, dt=0.1)
              for i in range(n)]
    _apply_attrs(agents, attrs)
    obstacles = [Obstacle(pos=list(p), shape_dict={'shape': 'sphere', 'feature': r}, id=i)
                 for i, (p, r) in enumerate(obstacles_spec)]
    env = env_mod.MACAEnv()
    with contextlib.redirect_stdout(io.StringIO()):
        env.set_agents(agents, obstacles=obstacles)
    REC = Recorder(n)
    rec = {k: [] for k in ('step', 'pos', 'vel', 'heading', 'flags', 'total_dist', 'goal', 'perm', 'vpref', 'called',
                           'nbr_valid', 'nbr_n', 'nbr_id', 'nbr_kind', 'nbr_dsq', 'n_suit', 'fallback', 'plane_fail',
                           'lp4', 'vpost', 'action', 'coll_after_policy', 'pos_after', 'vel_after', 'heading_after',
                           'flags_after', 'total_dist_after', 'perm_after')}
    # capture all_actions rows: wrap update_velocitie (mampenv.py:83)
    actions = np.zeros((n, 7), np.float32)
    coll_after = np.zeros(n, np.uint8)
    orig_upd = env_mod.update_velocitie
    first_upd = [True]

    def upd(agent, action):
        if first_upd[0]:
            # the policy loop (mampenv.py:34-40) has finished for every agent at this point
            first_upd[0] = False
            for a in agents:
                coll_after[a.id] = 1 if a.is_collision else 0
        actions[agent.id] = action
        return orig_upd(agent, action)
    env_mod.update_velocitie = upd
    t0 = time.time()
    done_step = -1
    for step in range(max_steps):
        want = (step < record_first) or (step % record_every == 0)
        REC.reset()
        first_upd[0] = True
        pre = _snapshot(agents)
        perm = np.array(env.kdTree.agentIDs, np.int32)
        with contextlib.redirect_stdout(io.StringIO()):
            done = env.step({})
        post = _snapshot(agents)
        if want:
            rec['step'].append(step)
            for k, v in zip(('pos', 'vel', 'heading', 'flags', 'total_dist', 'goal'), pre):
                rec[k].append(v)
            rec['perm'].append(perm)
            for k in ('vpref', 'called', 'nbr_valid', 'nbr_n', 'nbr_id', 'nbr_kind', 'nbr_dsq', 'n_suit', 'fallback',
                      'plane_fail', 'lp4', 'vpost'):
                rec[k].append(getattr(REC, k).copy())
            rec['action'].append(actions.copy())
            rec['coll_after_policy'].append(coll_after.copy())
            rec['pos_after'].append(post[0])
            rec['vel_after'].append(post[1])
            rec['heading_after'].append(post[2])
            rec['flags_after'].append(post[3])
            rec['total_dist_after'].append(post[4])
            rec['perm_after'].append(np.array(env.kdTree.agentIDs, np.int32))
        if done:
            done_step = step
            break
    env_mod.update_velocitie = orig_upd
    out = {k: np.array(v) for k, v in rec.items()}
    out.update(dict(
        name=name, n_steps_run=step + 1, done_step=done_step,
        start=np.array([np.asarray(p, dtype=np.float64) for p in pos]), goal6=np.array([np.asarray(g, dtype=np.float64) for g in goal]),
        radius=np.array(radius_l), pref_speed=np.array(ps_l), policy=np.array(policy_ids, np.uint8),
        max_run_dist=np.array([a.max_run_dist for a in agents]),
        obs_pos=np.array([p for p, _ in obstacles_spec], dtype=np.float64).reshape(-1, 3),
        obs_radius=np.array([r for _, r in obstacles_spec], dtype=np.float64),
        numpy_version=np.__version__, ))
    if attrs:
        out.update(_attr_arrays(agents))
    path = os.path.join(outdir, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: {step + 1} steps ({len(rec["step"])} recorded), done_step={done_step}, '
          f'{os.path.getsize(path) / 1e3:.0f} kB, {time.time() - t0:.1f} s', flush=True)
    return agents


def single_step_cluster(agent_mod, env_mod, classes, name, n, box, policy_id, seed, n_obs=0, min_sep=0.0,
                        outdir='tests/golden', attrs=None):
    """Hand-built dense state (F5): random positions in a cube, random float32 velocities (so the
    'first step' branch is not taken), one env.step.  Exercises >16 in range (agent.py:87-99 quirk),
    collisions (agent.py:82-85) and the no-suitable-candidate fallback (scaPolicy.py:224-238)."""
    global REC
    from mamp.agents.obstacle import Obstacle
    rng = np.random.default_rng(seed)
    pts = []
    while len(pts) < n:
        p = rng.uniform(-box / 2, box / 2, 3) + np.array([0.0, 0.0, 20.0])
        if min_sep > 0 and any(np.linalg.norm(p - q) < min_sep for q in pts):
            continue
        pts.append(p)
    pos = [list(p) + [float(rng.uniform(0, 2 * np.pi)), 0.0, 0.0] for p in pts]
    goal = [list(-np.array(p[:3]) + np.array([0, 0, 40.0])) + [0.0, 0.0, 0.0] for p in pos]
    agents = [agent_mod.Agent(start_pos=pos[i], goal_pos=goal[i], vel=[0.0, 0.0, 0.0], radius=0.5, pref_speed=1.0,
                              policy=classes[policy_id], id=i, dt=0.1) for i in range(n)]
    _apply_attrs(agents, attrs)
    for a in agents:
        v = rng.normal(size=3)
        v = v / np.linalg.norm(v) * rng.uniform(0.4, 1.0)
        a.vel_global_frame = v.astype(np.float32)
        if policy_id in (POL_SCA, POL_RVO_DUBINS):
            # keep the Dubins tracker out of this fixture: pre-mark as planned with an empty path so
            # compute_v_pref (scaPolicy.py:290-319) falls back to goal - pos when off-track
            pass
    obstacles_spec = [(list(rng.uniform(-box / 2, box / 2, 3) + np.array([0, 0, 20.0])), 1.0) for _ in range(n_obs)]
    obstacles = [Obstacle(pos=list(p), shape_dict={'shape': 'sphere', 'feature': r}, id=i)
                 for i, (p, r) in enumerate(obstacles_spec)]
    env = env_mod.MACAEnv()
    with contextlib.redirect_stdout(io.StringIO()):
        env.set_agents(agents, obstacles=obstacles)
    REC = Recorder(n)
    actions = np.zeros((n, 7), np.float32)
    coll_after = np.zeros(n, np.uint8)
    orig_upd = env_mod.update_velocitie
    first_upd = [True]

    def upd(agent, action):
        if first_upd[0]:
            first_upd[0] = False
            for a in agents:
                coll_after[a.id] = 1 if a.is_collision else 0
        actions[agent.id] = action
        return orig_upd(agent, action)
    env_mod.update_velocitie = upd
    pre = _snapshot(agents)
    perm = np.array(env.kdTree.agentIDs, np.int32)
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        env.step({})
    env_mod.update_velocitie = orig_upd
    post = _snapshot(agents)
    out = dict(name=name, step=np.array([0]), n_steps_run=1, done_step=-1)
    for k, v in zip(('pos', 'vel', 'heading', 'flags', 'total_dist', 'goal'), pre):
        out[k] = v[None]
    out['perm'] = perm[None]
    for k in ('vpref', 'called', 'nbr_valid', 'nbr_n', 'nbr_id', 'nbr_kind', 'nbr_dsq', 'n_suit', 'fallback',
              'plane_fail', 'lp4', 'vpost'):
        out[k] = getattr(REC, k).copy()[None]
    out['action'] = actions[None]
    out['coll_after_policy'] = coll_after[None]
    for k, v in zip(('pos_after', 'vel_after', 'heading_after', 'flags_after', 'total_dist_after'), post[:5]):
        out[k] = v[None]
    out['perm_after'] = np.array(env.kdTree.agentIDs, np.int32)[None]
    out.update(dict(start=np.array(pos, dtype=np.float64), goal6=np.array(goal, dtype=np.float64), radius=np.full(n, 0.5),
                    pref_speed=np.full(n, 1.0), policy=np.full(n, policy_id, np.uint8),
                    max_run_dist=np.array([a.max_run_dist for a in agents]),
                    obs_pos=np.array([p for p, _ in obstacles_spec], dtype=np.float64).reshape(-1, 3),
                    obs_radius=np.array([r for _, r in obstacles_spec], dtype=np.float64), numpy_version=np.__version__))
    if attrs:
        out.update(_attr_arrays(agents))
    path = os.path.join(outdir, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: n={n} nbr_n max={REC.nbr_n.max()} fallback={int(REC.fallback.sum())} '
          f'collisions={int(coll_after.sum())} lp4={int(REC.lp4.sum())} {time.time() - t0:.1f} s', flush=True)


def single_step_random(agent_mod, env_mod, classes, name, seed, outdir='tests/golden', attrs=None):
    """Fuzz fixtures (F12): a random small scene -- all six policies mixed, radii and preferred speeds of several sizes,
    obstacles that may overlap agents, agents done from the start (at goal / collided / timed out), agents at rest
    (bootstrap branch), goals straight above the start (is_zAxis), dense or sparse -- and one env.step of the reference."""
    global REC
    from mamp.agents.obstacle import Obstacle
    rng = np.random.default_rng(seed)
    n = int(rng.choice([3, 9, 17, 30, 45]))
    m = int(rng.choice([0, 2, 10]))
    side = float(rng.choice([3.0, 6.0, 15.0]))
    xyz = rng.uniform(-side, side, (n, 3))
    xyz[:, 2] = np.abs(xyz[:, 2]) + float(rng.choice([0.0, 1.0, 20.0]))
    g = rng.uniform(-side, side, (n, 3))
    g[:, 2] = np.abs(g[:, 2]) + 1.0
    if rng.random() < 0.4:
        g[: n // 2, :2] = xyz[: n // 2, :2]                               # is_zAxis agents (scaPolicy.py:188-190)
    pos = [list(map(float, xyz[i])) + [float(rng.uniform(0, 2 * np.pi)), float(rng.uniform(-0.4, 0.4)), 0.0] for i in range(n)]
    goal = [list(map(float, g[i])) + [float(rng.uniform(0, 2 * np.pi)), 0.0, 0.0] for i in range(n)]
    policy_ids = [int(x) for x in rng.integers(0, 6, n)]
    radius_l = [float(x) for x in rng.choice([0.3, 0.5, 1.0], n)]
    ps_l = [float(x) for x in rng.choice([1.0, 1.0, 0.8, 1.5], n)]
    agents = [agent_mod.Agent(start_pos=pos[i], goal_pos=goal[i], vel=[0.0, 0.0, 0.0], radius=radius_l[i], pref_speed=ps_l[i],
                              policy=classes[policy_ids[i]], id=i, dt=0.1) for i in range(n)]
    _apply_attrs(agents, attrs)
    rest = rng.random(n) < 0.2
    for a in agents:
        v = rng.normal(size=3)
        v = v / np.linalg.norm(v) * rng.uniform(0.1, 1.0)
        a.vel_global_frame = np.zeros(3, np.float32) if rest[a.id] else v.astype(np.float32)
        u = rng.random()
        if u < 0.05:
            a.is_at_goal = True
        elif u < 0.10:
            a.is_collision = True
        elif u < 0.13:
            a.is_out_of_max_time = True
    obstacles_spec = [(list(map(float, rng.uniform(-side, side, 3) * np.array([1, 1, 0.5]) + np.array([0, 0, side / 2]))),
                       float(rng.choice([0.2, 1.0, 2.0]))) for _ in range(m)]
    obstacles = [Obstacle(pos=list(p), shape_dict={'shape': 'sphere', 'feature': r}, id=i)
                 for i, (p, r) in enumerate(obstacles_spec)]
    env = env_mod.MACAEnv()
    with contextlib.redirect_stdout(io.StringIO()):
        env.set_agents(agents, obstacles=obstacles)
    REC = Recorder(n)
    actions = np.zeros((n, 7), np.float32)
    coll_after = np.zeros(n, np.uint8)
    orig_upd = env_mod.update_velocitie
    first_upd = [True]

    def upd(agent, action):
        if first_upd[0]:
            first_upd[0] = False
            for a in agents:
                coll_after[a.id] = 1 if a.is_collision else 0
        actions[agent.id] = action
        return orig_upd(agent, action)
    env_mod.update_velocitie = upd
    pre = _snapshot(agents)
    perm = np.array(env.kdTree.agentIDs, np.int32)
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        env.step({})
    env_mod.update_velocitie = orig_upd
    post = _snapshot(agents)
    out = dict(name=name, step=np.array([0]), n_steps_run=1, done_step=-1)
    for k, v in zip(('pos', 'vel', 'heading', 'flags', 'total_dist', 'goal'), pre):
        out[k] = v[None]
    out['perm'] = perm[None]
    for k in ('vpref', 'called', 'nbr_valid', 'nbr_n', 'nbr_id', 'nbr_kind', 'nbr_dsq', 'n_suit', 'fallback',
              'plane_fail', 'lp4', 'vpost'):
        out[k] = getattr(REC, k).copy()[None]
    out['action'] = actions[None]
    out['coll_after_policy'] = coll_after[None]
    for k, v in zip(('pos_after', 'vel_after', 'heading_after', 'flags_after', 'total_dist_after'), post[:5]):
        out[k] = v[None]
    out['perm_after'] = np.array(env.kdTree.agentIDs, np.int32)[None]
    out.update(dict(start=np.array(pos, dtype=np.float64), goal6=np.array(goal, dtype=np.float64), radius=np.array(radius_l),
                    pref_speed=np.array(ps_l), policy=np.array(policy_ids, np.uint8),
                    max_run_dist=np.array([a.max_run_dist for a in agents]),
                    obs_pos=np.array([p for p, _ in obstacles_spec], dtype=np.float64).reshape(-1, 3),
                    obs_radius=np.array([r for _, r in obstacles_spec], dtype=np.float64), numpy_version=np.__version__))
    if attrs:
        out.update(_attr_arrays(agents))
    path = os.path.join(outdir, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: n={n} m={m} side={side} called={int(REC.called.sum())} nbr_n max={REC.nbr_n.max()} '
          f'fallback={int((REC.fallback > 0).sum())} collisions={int(coll_after.sum())} lp4={int((REC.lp4 > 0).sum())} '
          f'{time.time() - t0:.1f} s', flush=True)


def kat_tables(outdir='tests/golden'):
    """F8: per-function known-answer tables for the scalar helpers of mamp/util.py."""
    import mamp.util as u
    from mamp.policies import orca3dPolicyOfficial as off
    rng = np.random.default_rng(1234)
    n = 4000
    a = rng.uniform(-30, 30, (n, 3))
    b = rng.uniform(-30, 30, (n, 3))
    # edge: tiny separations, exact decimal grids
    a[:200] = np.round(a[:200], 2)
    b[:200] = np.round(b[:200], 2)
    b[200:300] = a[200:300] + rng.uniform(-1e-3, 1e-3, (100, 3))
    l3 = np.array([u.l3norm(a[i], b[i]) for i in range(n)])
    l3sq = np.array([float(u.l3normsq(a[i], b[i])) for i in range(n)])
    dist = np.array([u.distance(a[i], b[i]) for i in range(n)])
    v = rng.uniform(-1, 1, (n, 3))
    v[:50, 1] = 0.0
    v[50:100, 0] = 0.0
    phi = np.array([u.get_phi(v[i]) for i in range(n)])
    ang = rng.uniform(-20, 20, n)
    p2p = np.array([u.pi_2_pi(np.float64(x)) for x in ang])
    m2p = np.array([u.mod2pi(x) for x in ang])
    # l3norm with a float32 second operand (scaPolicy.py:128: l3norm(v, vA), vA float32)
    vf = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    l3_mixed = np.array([u.l3norm(v[i], vf[i]) for i in range(n)])
    l3_f32zero = np.array([u.l3norm(vf[i], [0, 0, 0]) for i in range(n)])
    # is_intersect (util.py:30-41)
    pA = rng.uniform(-5, 5, (n, 3))
    pB = pA + rng.uniform(-6, 6, (n, 3))
    R = rng.uniform(0.6, 2.5, n)
    vd = rng.uniform(-1.5, 1.5, (n, 3))
    isx = np.zeros(n, np.uint8)
    for i in range(n):
        try:
            isx[i] = 1 if u.is_intersect(pA[i], pB[i], R[i], vd[i]) else 0
        except ValueError:
            isx[i] = 2
    # satisfied_constraint (util.py:6-20)

    class A:
        pass
    sat = np.zeros(n, np.uint8)
    posz = rng.uniform(-0.05, 0.3, n)
    for i in range(n):
        ag = A()
        ag.vel_global_frame = vf[i]
        ag.pos_global_frame = np.array([0.0, 0.0, posz[i]])
        ag.timeStep = 0.1
        ag.max_heading_change = math.pi / 4
        sat[i] = 1 if u.satisfied_constraint(ag, v[i]) else 0
    # cartesian2spherical (util.py:44-55) and the ORCA-official private copy (orca3dPolicyOfficial.py:331)
    head = rng.uniform(-3.2, 3.2, (n, 3))
    vv = v.copy()
    vv[:20] *= 1e-4
    c2s = np.zeros((n, 7))
    c2s_off = np.zeros((n, 7))
    for i in range(n):
        ag = A()
        ag.heading_global_frame = head[i]
        c2s[i] = u.cartesian2spherical(ag, vv[i])
        c2s_off[i] = off.cartesian2spherical(ag, vv[i])
    # truncation int(x*1e5)/1e5 (scaPolicy.py:239)
    tr_in = np.concatenate([rng.uniform(-1.2, 1.2, n - 6), [0.3, 0.29999999999999993, -0.3, 1.0, -1.0, 0.0]])
    tr = np.array([int(x * 1e5) / 1e5 for x in tr_in])
    np.savez_compressed(os.path.join(outdir, 'F8_kat.npz'), a=a, b=b, l3=l3, l3sq=l3sq, dist=dist, v=v, phi=phi, ang=ang,
                        p2p=p2p, m2p=m2p, vf=vf, l3_mixed=l3_mixed, l3_f32zero=l3_f32zero, pA=pA, pB=pB, R=R, vd=vd,
                        isx=isx, sat=sat, posz=posz, head=head, vv=vv, c2s=c2s, c2s_off=c2s_off, tr_in=tr_in, tr=tr,
                        numpy_version=np.__version__)
    print('F8_kat written', flush=True)


def candidate_table(outdir='tests/golden'):
    """The 2x256 (+2x128) Fibonacci-sphere candidate table exactly as scaPolicy.py:195-200 builds it."""
    from math import sqrt, cos, sin, pi
    out = {}
    for num_N in (256, 128):
        rows = []
        for rad in np.arange(0.5, 1.0 + 0.03, 1.0 - 0.5):
            for nn in range(1, num_N + 1):
                z_n = (2 * nn - 1) / num_N - 1
                x_n = sqrt(1 - z_n ** 2) * cos(2 * pi * nn * ((sqrt(5.0) - 1.0) / 2.0))
                y_n = sqrt(1 - z_n ** 2) * sin(2 * pi * nn * ((sqrt(5.0) - 1.0) / 2.0))
                rows.append([rad * x_n, rad * y_n, rad * z_n])
        out[f'cand{num_N}'] = np.array(rows)
    np.savez_compressed(os.path.join(outdir, 'F0_candidates.npz'), **out)
    print('F0_candidates written', flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden'))
    ap.add_argument('--only', nargs='*', default=None)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    agent_mod, env_mod, mods, classes = _import_reference(args.ref)
    _install_wrappers(mods)
    import run_sca as rs
    import run_orca as ro

    def want(nm):
        return args.only is None or any(nm.startswith(o) for o in args.only)

    od = args.out
    if want('F0'):
        candidate_table(od)
    if want('F8'):
        kat_tables(od)
    # F1: BASELINE config 1 -- N=8 circle rad 10, SCA, whole episode (run_sca.py:17-30)
    if want('F1'):
        pos, goal = rs.set_circle_pos((0, 0), 10.0, 8)
        run_env_episode(agent_mod, env_mod, classes, 'F1_sca_circle8', pos, goal, [POL_SCA] * 8, [], 400, outdir=od)
    # F2: N=100 circle rad 18 (run_orca.py:16-34), 40 steps, every policy
    names = {POL_SCA: 'sca', POL_RVO: 'rvo', POL_SRVO: 'srvo', POL_ORCA: 'orca', POL_ORCA_LP: 'orcalp',
             POL_RVO_DUBINS: 'rvodubins'}
    for pid in (POL_ORCA_LP, POL_RVO, POL_SRVO, POL_ORCA, POL_SCA, POL_RVO_DUBINS):
        nm = f'F2_{names[pid]}_circle100'
        if want(nm):
            pos, goal, _ = ro.set_circle_pos(100)
            steps = 40 if pid not in (POL_RVO_DUBINS,) else 12
            run_env_episode(agent_mod, env_mod, classes, nm, pos, goal, [pid] * 100, [], steps, outdir=od)
    # F3: N=100 random cube / Fibonacci ball (run_orca.py:36-71), seeded
    for pid in (POL_ORCA_LP, POL_RVO, POL_SRVO, POL_ORCA):
        nm = f'F3_{names[pid]}_random100'
        if want(nm):
            random.seed(7)
            pos, goal, _ = ro.set_random_pos(100)
            run_env_episode(agent_mod, env_mod, classes, nm, pos, goal, [pid] * 100, [], 25, outdir=od)
    for pid in (POL_ORCA_LP, POL_SRVO):
        nm = f'F3_{names[pid]}_sphere100'
        if want(nm):
            pos, goal, _ = ro.set_sphere(100)
            run_env_episode(agent_mod, env_mod, classes, nm, pos, goal, [pid] * 100, [], 25, outdir=od)
    # F4: N=16 take-off/landing + 8 sphere obstacles (run_sca.py:53-81,139-150), SCA and mixed SCA/S-RVO3D
    obs = []
    for j in range(8):
        obs.append(([round(4.0 * np.cos(2 * j * np.pi / 8), 2), round(4.0 * np.sin(2 * j * np.pi / 8), 2), 5.0], 1.0))
    if want('F4_sca_takeoff16'):
        pos, goal = rs.set_takeoff_landing_pos(16)
        run_env_episode(agent_mod, env_mod, classes, 'F4_sca_takeoff16', pos, goal, [POL_SCA] * 16, obs, 400, outdir=od)
    if want('F4_mixed_takeoff16'):
        pos, goal = rs.set_takeoff_landing_pos(16)
        pol = [POL_SCA if i % 2 == 0 else POL_SRVO for i in range(16)]
        run_env_episode(agent_mod, env_mod, classes, 'F4_mixed_takeoff16', pos, goal, pol, obs, 400, outdir=od)
    if want('F4_sca_circle16_obs'):
        pos, goal = rs.set_circle_pos((0, 0), 10.0, 16)
        run_env_episode(agent_mod, env_mod, classes, 'F4_sca_circle16_obs', pos, goal, [POL_SCA] * 16, obs, 400,
                        record_every=3, record_first=5, outdir=od)
    # F5: dense clusters, one step, every solver
    for pid in (POL_RVO, POL_SRVO, POL_ORCA, POL_ORCA_LP):
        nm = f'F5_{names[pid]}_dense80'
        if want(nm):
            single_step_cluster(agent_mod, env_mod, classes, nm, 80, 12.0, pid, seed=11 + pid, n_obs=6, outdir=od)
        nm = f'F5_{names[pid]}_packed60'
        if want(nm):
            single_step_cluster(agent_mod, env_mod, classes, nm, 60, 9.0, pid, seed=31 + pid, n_obs=14, min_sep=1.05,
                                outdir=od)
    # F9: heterogeneous radii and preferred speeds, mixed policies, obstacles of different sizes
    if want('F9_hetero_mixed60'):
        random.seed(23)
        rng = np.random.default_rng(23)
        r = 9.0
        pos = [np.array([random.uniform(-r, r), random.uniform(-r, r), random.uniform(-r, r) + 30.0, random.uniform(0, 6.28), 0.0, 0.0]) for _ in range(60)]
        goal = [np.array([random.uniform(-r, r), random.uniform(-r, r), random.uniform(-r, r) + 30.0, 0.0, 0.0, 0.0]) for _ in range(60)]
        pol = [[POL_RVO, POL_SRVO, POL_ORCA, POL_ORCA_LP][i % 4] for i in range(60)]
        rad = rng.choice([0.3, 0.5, 0.8], 60)
        psp = rng.choice([0.8, 1.0, 1.5], 60)
        obs9 = [([4.0, 1.0, 31.0], 1.5), ([-5.0, -3.0, 27.0], 0.6), ([0.0, 6.0, 34.0], 2.5)]
        run_env_episode(agent_mod, env_mod, classes, 'F9_hetero_mixed60', pos, goal, pol, obs9, 30, radius=rad,
                        pref_speed=psp, outdir=od)
    # F10: exp3 "low altitude search" (run_sca.py:118,152-154): 16 drones among the 1491 r=0.2 spheres that
    # read_map.read_obstacle extracts from visualization/map/map.binvox -- an obstacle kd-tree with real depth,
    # obstacles filling the neighbour lists.  Only the obstacle list (data) enters the fixture.
    if want('F10_sca_exp3_map'):
        np.bool = bool                      # harness shim: read_map.py:19 uses the alias numpy 2 removed (asset loading only)
        from mamp.read_map import read_obstacle
        with contextlib.redirect_stdout(io.StringIO()):
            objs = read_obstacle(center=(35, 30), environ="exp3", obs_path=os.path.join(args.ref, 'visualization', 'map', 'map.binvox'))
        obs10 = [(list(map(float, o.pos_global_frame)), float(o.radius)) for o in objs]
        pos, goal = rs.spawn_n_drones(center=(35, 30), rad=10.0, drone_num=16, environment="exp3")
        pos = [[float(v) for v in p_] for p_ in pos]
        goal = [[float(v) for v in g_] for g_ in goal]
        run_env_episode(agent_mod, env_mod, classes, 'F10_sca_exp3_map', pos, goal, [POL_SCA] * 16, obs10, 160, outdir=od)
    # F12: fuzz -- random small scenes, one step each, everything mixed (found by fuzzing the HIP path against the oracle:
    # cases no hand-made fixture had, e.g. an agent that is at its goal AND inside an obstacle)
    for k in range(16):
        nm = f'F12_fuzz_{k:02d}'
        if want(nm):
            single_step_random(agent_mod, env_mod, classes, nm, seed=1200 + k, outdir=od)
    # F13: fuzz episodes for the v_pref tracker -- random starts / goals with pitched and yawed poses, take-off agents,
    # three preferred speeds, SCA and RVO3D+Dubins among the other policies, 40 steps each
    for k in range(4):
        nm = f'F13_fuzz_track_{k:02d}'
        if want(nm):
            rng = np.random.default_rng(1300 + k)
            n = int(rng.choice([8, 12, 16]))
            side = float(rng.choice([6.0, 12.0, 25.0]))
            xyz = rng.uniform(-side, side, (n, 3))
            xyz[:, 2] = np.abs(xyz[:, 2]) + float(rng.choice([0.0, 5.0]))
            g = rng.uniform(-side, side, (n, 3))
            g[:, 2] = np.abs(g[:, 2]) + 1.0
            g[: n // 3, :2] = xyz[: n // 3, :2]                                # take-off / landing agents (is_zAxis)
            pos = [list(map(float, xyz[i])) + [float(rng.uniform(0, 2 * np.pi)), float(rng.uniform(-0.5, 0.5)), 0.0] for i in range(n)]
            goal = [list(map(float, g[i])) + [float(rng.uniform(0, 2 * np.pi)), float(rng.uniform(-0.3, 0.3)), 0.0] for i in range(n)]
            pol = [int(x) for x in rng.choice([POL_SCA, POL_SCA, POL_RVO_DUBINS, POL_SRVO, POL_ORCA_LP], n)]
            psp = rng.choice([0.8, 1.0, 1.5], n)
            obs13 = [(list(map(float, rng.uniform(-side, side, 3) * np.array([1, 1, 0.5]) + np.array([0, 0, side / 2]))), 1.0) for _ in range(3)]
            run_env_episode(agent_mod, env_mod, classes, nm, pos, goal, pol, obs13, 40, pref_speed=psp, outdir=od)
    # F14: fuzz episodes -- dense random boxes, every policy but the two tracked ones, goals close by (arrivals, collisions
    # and time-outs happen within the 20 recorded steps), obstacles among the agents
    for k in range(3):
        nm = f'F14_fuzz_episode_{k:02d}'
        if want(nm):
            rng = np.random.default_rng(1400 + k)
            n = int(rng.choice([20, 30, 40]))
            side = float(rng.choice([4.0, 7.0]))
            xyz = rng.uniform(-side, side, (n, 3))
            xyz[:, 2] = np.abs(xyz[:, 2]) + 1.0
            g = xyz + rng.normal(0, 1.5, (n, 3))                               # short trips: many arrive
            g[:, 2] = np.abs(g[:, 2]) + 0.5
            pos = [list(map(float, xyz[i])) + [float(rng.uniform(0, 2 * np.pi)), 0.0, 0.0] for i in range(n)]
            goal = [list(map(float, g[i])) + [0.0, 0.0, 0.0] for i in range(n)]
            pol = [int(x) for x in rng.choice([POL_RVO, POL_SRVO, POL_ORCA, POL_ORCA_LP], n)]
            rad = rng.choice([0.3, 0.5], n)
            psp = rng.choice([0.8, 1.0, 1.5], n)
            obs14 = [(list(map(float, rng.uniform(-side, side, 3) * np.array([1, 1, 0.5]) + np.array([0, 0, side / 2]))),
                      float(rng.choice([0.3, 1.0]))) for _ in range(4)]
            run_env_episode(agent_mod, env_mod, classes, nm, pos, goal, pol, obs14, 20, radius=rad, pref_speed=psp, outdir=od)
    # F15: BASELINE config 2 itself -- N = 1024 circle, rad = 1.25 N / 2 pi (the arc spacing bench.py and SURVEY 8d use), SCA with
    # its Dubins tracker, steps 0-3 stepped by the reference (about ten minutes of Python: ~2000 Dubins plans of 412 m)
    if want('F15_sca_circle1024'):
        pos, goal = rs.set_circle_pos((0, 0), 1.25 * 1024 / (2.0 * math.pi), 1024)
        pos = [[float(v) for v in p_] for p_ in pos]
        goal = [[float(v) for v in g_] for g_ in goal]
        run_env_episode(agent_mod, env_mod, classes, 'F15_sca_circle1024', pos, goal, [POL_SCA] * 1024, [], 4, outdir=od)
    # F16: the parameter space the C-ABI exports (sca_params, sca_device_tracker_enable) -- the reference stepped with the solver attributes
    # changed after Agent.__init__ (agent.py:24-41), one value per scene for all agents.  Readers: scaPolicy.py:95,112,272,299-302,
    # util.py:8,17, orca3dPolicyOfficial.py:44,98,108, agent.py:87-99, mampenv.py:90-92 (dt_nominal).
    def _cube(rng, n, side, z0, min_sep):
        pts = []
        while len(pts) < n:
            q = rng.uniform(-side, side, 3) + np.array([0.0, 0.0, z0])
            if all(np.linalg.norm(q - r_) >= min_sep for r_ in pts):
                pts.append(q)
        return np.array(pts)

    if want('F16_params_nbr4_dense40'):
        # more than maxNeighbors = 4 objects inside neighborDist = 4.0 for most agents; all six policies; radii 0.3 / 0.8
        rng = np.random.default_rng(1601)
        n = 40
        xyz = _cube(rng, n, 4.0, 20.0, 1.9)
        g = -xyz + np.array([0.0, 0.0, 40.0])
        pos = [list(map(float, xyz[i])) + [float(rng.uniform(0, 2 * np.pi)), 0.0, 0.0] for i in range(n)]
        goal = [list(map(float, g[i])) + [float(rng.uniform(0, 2 * np.pi)), 0.0, 0.0] for i in range(n)]
        pol = [i % 6 for i in range(n)]
        rad = rng.choice([0.3, 0.8], n)
        obs = [([0.5, -0.5, 20.0], 0.7), ([3.0, 3.0, 22.0], 0.4)]
        run_env_episode(agent_mod, env_mod, classes, 'F16_params_nbr4_dense40', pos, goal, pol, obs, 16, radius=rad, outdir=od,
                        attrs=dict(maxNeighbors=4, neighborDist=4.0))
    if want('F16_params_nbr8_far60'):
        # neighborDist = 15 (a wider search than the default's 10) with maxNeighbors = 8: circle of 60, every policy
        pos, goal = rs.set_circle_pos((0, 0), 12.0, 60)
        pos = [[float(v) for v in p_] for p_ in pos]
        goal = [[float(v) for v in g_] for g_ in goal]
        pol = [i % 6 for i in range(60)]
        run_env_episode(agent_mod, env_mod, classes, 'F16_params_nbr8_far60', pos, goal, pol, [], 14, outdir=od,
                        attrs=dict(maxNeighbors=8, neighborDist=15.0))
    if want('F16_params_orca_h3_v15'):
        # timeHorizon = 3, maxSpeed = 1.5, pref_speed = 1.2: both ORCA policies (planes, LP3 / LP4 with the larger speed ball) beside RVO / S-RVO
        rng = np.random.default_rng(1603)
        n = 60
        xyz = _cube(rng, n, 6.0, 25.0, 1.3)
        g = -xyz + np.array([0.0, 0.0, 50.0])
        pos = [list(map(float, xyz[i])) + [float(rng.uniform(0, 2 * np.pi)), 0.0, 0.0] for i in range(n)]
        goal = [list(map(float, g[i])) + [0.0, 0.0, 0.0] for i in range(n)]
        pol = [[POL_ORCA_LP, POL_ORCA, POL_ORCA_LP, POL_RVO, POL_ORCA_LP, POL_SRVO][i % 6] for i in range(n)]
        run_env_episode(agent_mod, env_mod, classes, 'F16_params_orca_h3_v15', pos, goal, pol, [([0.0, 0.0, 25.0], 1.0)], 30,
                        pref_speed=1.2, outdir=od, attrs=dict(timeHorizon=3.0, maxSpeed=1.5))
    if want('F16_params_pitch30'):
        # pitchlims = +-pi/6 for the planner AND max_heading_change = pi/6 for the posture filter: take-off / landing scene with
        # obstacles, SCA / RVO3D+Dubins (tracked) beside RVO / S-RVO
        obs = [([round(4.0 * np.cos(2 * j * np.pi / 8), 2), round(4.0 * np.sin(2 * j * np.pi / 8), 2), 5.0], 1.0) for j in range(8)]
        pos, goal = rs.set_takeoff_landing_pos(16)
        pol = [[POL_SCA, POL_RVO_DUBINS, POL_SCA, POL_SRVO, POL_SCA, POL_RVO][i % 6] for i in range(16)]
        run_env_episode(agent_mod, env_mod, classes, 'F16_params_pitch30', pos, goal, pol, obs, 60, outdir=od,
                        attrs=dict(pitchlims=(-math.pi / 6, math.pi / 6), min_heading_change=-math.pi / 6, max_heading_change=math.pi / 6))
    if want('F16_params_turn3_sca16'):
        # turning_radius = 3.0 (k = 9, condition_dist at 6, scaPolicy.py:272,302): SCA circle of 16 with its tracker, radii 0.3 / 0.8, 40 steps
        rng = np.random.default_rng(1605)
        pos, goal = rs.set_circle_pos((0, 0), 10.0, 16)
        pos = [[float(v) for v in p_] for p_ in pos]
        goal = [[float(v) for v in g_] for g_ in goal]
        run_env_episode(agent_mod, env_mod, classes, 'F16_params_turn3_sca16', pos, goal, [POL_SCA] * 16, [], 40,
                        radius=rng.choice([0.3, 0.8], 16), outdir=od, attrs=dict(turning_radius=3.0))
    if want('F16_params_turn08_takeoff12'):
        # turning_radius = 0.8 with take-off agents (is_zAxis: condition_dist compares the nearest neighbour with 1.6) and pitchlims (-0.5, 0.9)
        pos, goal = rs.set_takeoff_landing_pos(12)
        pol = [POL_SCA if i % 3 else POL_RVO_DUBINS for i in range(12)]
        run_env_episode(agent_mod, env_mod, classes, 'F16_params_turn08_takeoff12', pos, goal, pol, [([0.0, 0.0, 6.0], 1.2)], 50, outdir=od,
                        attrs=dict(turning_radius=0.8, pitchlims=(-0.5, 0.9), neighborDist=2.5))
    if want('F16_params_timestep02'):
        # timeStep = 0.2 while the integrator keeps dt_nominal = 0.1 (util.py:8, orca3dPolicyOfficial.py:98 against mampenv.py:90-92)
        rng = np.random.default_rng(1607)
        n = 36
        xyz = _cube(rng, n, 4.0, 3.0, 1.2)
        xyz[:, 2] = np.abs(xyz[:, 2] - 3.0) * 0.3                                  # near the ground: the next_pA[2] >= 0 test bites
        g = -xyz + np.array([0.0, 0.0, 6.0])
        pos = [list(map(float, xyz[i])) + [float(rng.uniform(0, 2 * np.pi)), 0.0, 0.0] for i in range(n)]
        goal = [list(map(float, g[i])) + [0.0, 0.0, 0.0] for i in range(n)]
        pol = [[POL_RVO, POL_SRVO, POL_ORCA, POL_ORCA_LP][i % 4] for i in range(n)]
        run_env_episode(agent_mod, env_mod, classes, 'F16_params_timestep02', pos, goal, pol, [], 20, outdir=od, attrs=dict(timeStep=0.2))
    if want('F16_params_dt005'):
        # dt_nominal = 0.05 and timeStep = 0.05: half-length steps in the integrator and in the constraints
        rng = np.random.default_rng(1608)
        n = 24
        xyz = _cube(rng, n, 3.0, 10.0, 1.2)
        g = -xyz + np.array([0.0, 0.0, 20.0])
        pos = [list(map(float, xyz[i])) + [float(rng.uniform(0, 2 * np.pi)), 0.0, 0.0] for i in range(n)]
        goal = [list(map(float, g[i])) + [0.0, 0.0, 0.0] for i in range(n)]
        pol = [[POL_SCA, POL_RVO, POL_SRVO, POL_ORCA, POL_ORCA_LP, POL_RVO_DUBINS][i % 6] for i in range(n)]
        run_env_episode(agent_mod, env_mod, classes, 'F16_params_dt005', pos, goal, pol, [], 24, outdir=od,
                        attrs=dict(timeStep=0.05, dt_nominal=0.05))
    if want('F16_params_all_mixed48'):
        # everything off its default at once, obstacles among the agents, three preferred speeds
        rng = np.random.default_rng(1609)
        n = 48
        xyz = _cube(rng, n, 5.0, 12.0, 1.7)
        g = -xyz + np.array([0.0, 0.0, 24.0])
        g[: n // 6, :2] = xyz[: n // 6, :2]                                         # a few take-off agents
        pos = [list(map(float, xyz[i])) + [float(rng.uniform(0, 2 * np.pi)), float(rng.uniform(-0.3, 0.3)), 0.0] for i in range(n)]
        goal = [list(map(float, g[i])) + [float(rng.uniform(0, 2 * np.pi)), 0.0, 0.0] for i in range(n)]
        pol = [int(x) for x in rng.integers(0, 6, n)]
        obs = [(list(map(float, rng.uniform(-5, 5, 3) + np.array([0, 0, 12.0]))), float(rng.choice([0.4, 1.0]))) for _ in range(5)]
        run_env_episode(agent_mod, env_mod, classes, 'F16_params_all_mixed48', pos, goal, pol, obs, 30, radius=rng.choice([0.3, 0.5, 0.8], n),
                        pref_speed=rng.choice([0.8, 1.0, 1.5], n), outdir=od,
                        attrs=dict(maxNeighbors=6, neighborDist=7.5, timeHorizon=5.0, maxSpeed=2.0, min_heading_change=-0.6, max_heading_change=0.6,
                                   pitchlims=(-0.5, 0.7), turning_radius=2.0, timeStep=0.1))
    # F16 fuzz: the F12 scenes (one step, everything mixed, agents done from the start, obstacles overlapping agents) under random parameter sets
    for k in range(8):
        nm = f'F16_params_fuzz_{k:02d}'
        if want(nm):
            rng = np.random.default_rng(1650 + k)
            mhc = float(rng.choice([math.pi / 4, math.pi / 6, 0.3, 1.2, math.pi / 2]))
            at = dict(maxNeighbors=int(rng.choice([1, 2, 4, 8, 12, 16])), neighborDist=float(rng.choice([1.5, 2.5, 4.0, 10.0, 15.0, 30.0])),
                      timeHorizon=float(rng.choice([1.0, 3.0, 10.0, 20.0])), maxSpeed=float(rng.choice([0.7, 1.0, 1.5, 3.0])),
                      min_heading_change=-mhc, max_heading_change=mhc, turning_radius=float(rng.choice([0.8, 1.5, 3.0, 10.0])),
                      pitchlims=(-float(rng.choice([0.3, math.pi / 6, math.pi / 4, 1.0])), float(rng.choice([0.3, math.pi / 6, math.pi / 4, 1.0]))),
                      timeStep=float(rng.choice([0.05, 0.1, 0.2])))
            single_step_random(agent_mod, env_mod, classes, nm, seed=1660 + k, outdir=od, attrs=at)
    # F16 dense one-step clusters (the F5 scenes) with a short list and a short range: > maxNeighbors in range, collisions, fallbacks
    for pid in (POL_RVO, POL_SRVO, POL_ORCA, POL_ORCA_LP):
        nm = f'F16_params_{names[pid]}_packed60'
        if want(nm):
            single_step_cluster(agent_mod, env_mod, classes, nm, 60, 9.0, pid, seed=31 + pid, n_obs=14, min_sep=1.05, outdir=od,
                                attrs=dict(maxNeighbors=5, neighborDist=3.0, timeHorizon=2.0, maxSpeed=1.3, max_heading_change=1.0))
    # F17: the attributes PER AGENT, as the reference keeps them (agent.py:24-41 are per-object; every policy reads its own agent's) -- each agent
    # draws its own maxNeighbors / neighborDist / timeStep / timeHorizon / maxSpeed / max_heading_change / dt_nominal (turning_radius and pitchlims
    # stay uniform: the tracker's context takes one value of each)
    def _hetero(seed):
        def draw(n):
            rng = np.random.default_rng(seed)
            mhc = rng.choice([0.5, math.pi / 4, 1.0, math.pi / 2], n)
            return dict(maxNeighbors=rng.choice([2, 5, 9, 16], n), neighborDist=rng.choice([3.0, 6.5, 10.0, 15.0], n), timeStep=rng.choice([0.05, 0.1, 0.2], n),
                        timeHorizon=rng.choice([2.0, 5.0, 10.0], n), maxSpeed=rng.choice([0.8, 1.0, 1.5, 2.5], n), min_heading_change=-mhc, max_heading_change=mhc,
                        dt_nominal=rng.choice([0.05, 0.1], n))
        return draw
    if want('F17_hetero_mixed48'):
        rng = np.random.default_rng(1701)
        n = 48
        xyz = _cube(rng, n, 5.0, 12.0, 1.7)
        g = -xyz + np.array([0.0, 0.0, 24.0])
        g[: n // 6, :2] = xyz[: n // 6, :2]
        pos = [list(map(float, xyz[i])) + [float(rng.uniform(0, 2 * np.pi)), float(rng.uniform(-0.3, 0.3)), 0.0] for i in range(n)]
        goal = [list(map(float, g[i])) + [float(rng.uniform(0, 2 * np.pi)), 0.0, 0.0] for i in range(n)]
        pol = [int(x) for x in rng.integers(0, 6, n)]
        obs = [(list(map(float, rng.uniform(-5, 5, 3) + np.array([0, 0, 12.0]))), float(rng.choice([0.4, 1.0]))) for _ in range(5)]
        run_env_episode(agent_mod, env_mod, classes, 'F17_hetero_mixed48', pos, goal, pol, obs, 30, radius=rng.choice([0.3, 0.5, 0.8], n),
                        pref_speed=rng.choice([0.8, 1.0, 1.5], n), outdir=od, attrs=_hetero(1711))
    if want('F17_hetero_circle60'):
        pos, goal = rs.set_circle_pos((0, 0), 12.0, 60)
        pos = [[float(v) for v in p_] for p_ in pos]
        goal = [[float(v) for v in g_] for g_ in goal]
        run_env_episode(agent_mod, env_mod, classes, 'F17_hetero_circle60', pos, goal, [i % 6 for i in range(60)], [], 18, outdir=od, attrs=_hetero(1712))
    if want('F17_hetero_dense40'):
        rng = np.random.default_rng(1703)
        n = 40
        xyz = _cube(rng, n, 4.0, 20.0, 1.9)
        g = -xyz + np.array([0.0, 0.0, 40.0])
        pos = [list(map(float, xyz[i])) + [float(rng.uniform(0, 2 * np.pi)), 0.0, 0.0] for i in range(n)]
        goal = [list(map(float, g[i])) + [0.0, 0.0, 0.0] for i in range(n)]
        run_env_episode(agent_mod, env_mod, classes, 'F17_hetero_dense40', pos, goal, [[POL_RVO, POL_SRVO, POL_ORCA, POL_ORCA_LP][i % 4] for i in range(n)],
                        [([0.5, -0.5, 20.0], 0.7)], 16, radius=rng.choice([0.3, 0.8], n), outdir=od, attrs=_hetero(1713))
    for k in range(6):
        nm = f'F17_hetero_fuzz_{k:02d}'
        if want(nm):
            single_step_random(agent_mod, env_mod, classes, nm, seed=1760 + k, outdir=od, attrs=_hetero(1770 + k))
    # F18: the PLANNER's attributes per agent -- turning_radius and pitchlims differ from agent to agent (three radii, three pairs of limits),
    # SCA and RVO3D+Dubins agents among untracked ones; the solver attributes at their defaults in the first two, drawn per agent in the third
    def _hetero_track(seed, with_solver):
        def draw(n):
            rng = np.random.default_rng(seed)
            pl = np.array([(-math.pi / 4, math.pi / 4), (-math.pi / 6, math.pi / 6), (-0.5, 0.9)])[rng.integers(0, 3, n)]
            at = dict(turning_radius=rng.choice([0.8, 1.5, 3.0], n), pitchlims=pl)
            if with_solver:
                at.update(_hetero(seed + 1)(n))
            return at
        return draw
    if want('F18_hetero_track_circle24'):
        pos, goal = rs.set_circle_pos((0, 0), 12.0, 24)
        pos = [[float(v) for v in p_] for p_ in pos]
        goal = [[float(v) for v in g_] for g_ in goal]
        pol = [[POL_SCA, POL_SCA, POL_RVO_DUBINS, POL_SRVO][i % 4] for i in range(24)]
        run_env_episode(agent_mod, env_mod, classes, 'F18_hetero_track_circle24', pos, goal, pol, [], 50, outdir=od, attrs=_hetero_track(1801, False))
    # every tracked agent its OWN (turning_radius, pitchlims): 30 distinct settings -- more than the 16 classes the device tracker's
    # class form holds, so its per-agent form runs
    if want('F18_hetero_track_circle30_each'):
        def _each(n):
            rng = np.random.default_rng(1805)
            lo = -(0.35 + 0.4 * rng.random(n))
            hi = 0.35 + 0.5 * rng.random(n)
            return dict(turning_radius=np.round(0.8 + 2.2 * rng.random(n), 3), pitchlims=np.stack([lo, hi], 1))
        pos, goal = rs.set_circle_pos((0, 0), 14.0, 30)
        pos = [[float(v) for v in p_] for p_ in pos]
        goal = [[float(v) for v in g_] for g_ in goal]
        pol = [POL_SCA if i % 4 else POL_RVO_DUBINS for i in range(30)]
        run_env_episode(agent_mod, env_mod, classes, 'F18_hetero_track_circle30_each', pos, goal, pol, [], 45, outdir=od, attrs=_each)
    if want('F18_hetero_track_takeoff16'):
        obs = [([round(4.0 * np.cos(2 * j * np.pi / 8), 2), round(4.0 * np.sin(2 * j * np.pi / 8), 2), 5.0], 1.0) for j in range(8)]
        pos, goal = rs.set_takeoff_landing_pos(16)
        pol = [POL_SCA if i % 3 else POL_RVO_DUBINS for i in range(16)]
        run_env_episode(agent_mod, env_mod, classes, 'F18_hetero_track_takeoff16', pos, goal, pol, obs, 60, outdir=od, attrs=_hetero_track(1802, False))
    if want('F18_hetero_track_mixed36'):
        rng = np.random.default_rng(1803)
        n = 36
        xyz = _cube(rng, n, 5.0, 12.0, 1.7)
        g = -xyz + np.array([0.0, 0.0, 24.0])
        g[: n // 6, :2] = xyz[: n // 6, :2]
        pos = [list(map(float, xyz[i])) + [float(rng.uniform(0, 2 * np.pi)), float(rng.uniform(-0.3, 0.3)), 0.0] for i in range(n)]
        goal = [list(map(float, g[i])) + [float(rng.uniform(0, 2 * np.pi)), 0.0, 0.0] for i in range(n)]
        pol = [int(x) for x in rng.choice([POL_SCA, POL_SCA, POL_RVO_DUBINS, POL_ORCA_LP, POL_RVO], n)]
        run_env_episode(agent_mod, env_mod, classes, 'F18_hetero_track_mixed36', pos, goal, pol, [([1.0, 0.0, 12.0], 0.8)], 30, radius=rng.choice([0.3, 0.5], n),
                        outdir=od, attrs=_hetero_track(1804, True))
    # F6: ORCA-official N=100 circle, long run, every 10th step (LP4 coverage)
    if want('F6_orcalp_circle100_long'):
        pos, goal, _ = ro.set_circle_pos(100)
        run_env_episode(agent_mod, env_mod, classes, 'F6_orcalp_circle100_long', pos, goal, [POL_ORCA_LP] * 100, [],
                        600, record_every=10, record_first=0, outdir=od)


if __name__ == '__main__':
    main()
