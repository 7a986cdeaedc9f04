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
