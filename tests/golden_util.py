"""Helpers shared by the golden-vector tests: fixture loading and input assembly."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
POL_SCA, POL_RVO, POL_SRVO, POL_ORCA, POL_ORCA_LP, POL_RVO_DUBINS = 0, 1, 2, 3, 4, 5


def episode_fixtures():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'F[1-69]*_*.npz'))
                  if 'episode_log' not in p)       # F11_episode_log_*: different schema (tests/test_episode_log.py)


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))


def static_inputs(fx):
    """Per-agent constants derived from the fixture the way the host code derives them."""
    start = fx['start']
    goal6 = fx['goal6']
    p0pA = goal6[:, :3] - start[:, :3]
    zaxis = ((np.abs(p0pA[:, 0]) <= 1e-5) & (np.abs(p0pA[:, 1]) <= 1e-5)).astype(np.uint8)   # scaPolicy.py:188-189
    policy = fx['policy'].astype(np.uint8)
    vpref_mode = np.isin(policy, (POL_SCA, POL_RVO_DUBINS)).astype(np.uint8)
    return dict(zaxis=zaxis, policy=policy, vpref_mode=vpref_mode, radius=fx['radius'], pref_speed=fx['pref_speed'],
                obs_pos=fx['obs_pos'], obs_radius=fx['obs_radius'], max_run_dist=fx['max_run_dist'])


_ATTR_TO_PARAM = dict(neighborDist='neighbor_dist', maxNeighbors='max_neighbors', timeStep='time_step', timeHorizon='time_horizon',
                      maxSpeed='max_speed', max_heading_change='max_heading_change', dt_nominal='dt_nominal')


def fixture_params(fx):
    """The solver parameters a fixture was recorded under (F16: `attr_*` arrays, the reference's Agent attributes read back after the
    generator changed them, tools/gen_golden.py::_apply_attrs).  Returns (params, tracker): `params` in sca_params / oracle.set_params
    names, `tracker` = dict(turning_radius, pitchlims) for sca_device_tracker_enable / DubinsTracker.  Both empty for fixtures recorded at
    the reference's defaults (agent.py:24-41).  One value per scene: sca_params is per context."""
    params, tracker = {}, {}
    for attr, name in _ATTR_TO_PARAM.items():
        if 'attr_' + attr in fx:
            v = fx['attr_' + attr]
            if not (v == v[0]).all():
                continue                                   # F17: this attribute differs from agent to agent -- fixture_agent_params
            params[name] = int(v[0]) if name == 'max_neighbors' else float(v[0])
    if 'attr_turning_radius' in fx:
        # (F18: the planner's attributes differ from agent to agent -- the first agent's here, everybody's own in fixture_tracker_agent_params)
        tracker = dict(turning_radius=float(fx['attr_turning_radius'][0]),
                       pitchlims=(float(fx['attr_pitch_lo'][0]), float(fx['attr_pitch_hi'][0])))
    return params, tracker


def fixture_tracker_agent_params(fx):
    """F18: agent.turning_radius / agent.pitchlims per agent where they differ, in sca_device_tracker_set_agent_params /
    DubinsTracker.set_agent_params names; {} otherwise."""
    if 'attr_turning_radius' not in fx:
        return {}
    keys = dict(turning_radius='attr_turning_radius', pitch_lo='attr_pitch_lo', pitch_hi='attr_pitch_hi')
    if all((fx[v] == fx[v][0]).all() for v in keys.values()):
        return {}
    return {k: fx[v].astype(np.float64) for k, v in keys.items()}


def fixture_agent_params(fx):
    """F17: the attributes that differ from agent to agent, as arrays in sca_set_agent_params / oracle.set_agent_params names ({} for
    every other fixture).  The reference keeps all of them per Agent object (agent.py:24-41)."""
    out = {}
    for attr, name in _ATTR_TO_PARAM.items():
        if 'attr_' + attr in fx:
            v = fx['attr_' + attr]
            if not (v == v[0]).all():
                out[name] = v.astype(np.int32) if name == 'max_neighbors' else v.astype(np.float64)
    return out


def hetero_fixtures():
    return [n for n in episode_fixtures() if n.startswith(('F17_hetero', 'F18_hetero'))]


def param_fixtures():
    return [n for n in episode_fixtures() if n.startswith('F16_params')]


def tracked_param_fixtures():
    """F16 scenes with SCA / RVO3D+Dubins agents: their recorded v_pref is the reference's Dubins tracker at the scene's
    turning_radius / pitchlims / neighborDist."""
    return [n for n in param_fixtures() + hetero_fixtures() if np.isin(load(n)['policy'], (POL_SCA, POL_RVO_DUBINS)).any()]
