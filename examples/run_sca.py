#!/usr/bin/env python3
"""The reference's run_example/run_sca.py, on the MI355X path: same scenario builders, same `while env.step(...)` loop, same
log files -- only the imports differ (sca_amd.env instead of mamp.*).

    python examples/run_sca.py                     # circle, 16 drones, 8 obstacle spheres (the reference's default run)
    python examples/run_sca.py --scenario takeoff  # exp2: take-off / landing
    python examples/run_sca.py --scenario exp3 --map tests/golden/exp3_map.binvox   # low-altitude search among 1491 spheres
    python examples/run_sca.py --scenario circle --agents 2000 --no-obstacles --policy orca
"""
import argparse
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sca_amd import env as E, metrics, read_map, scenarios, solver as S, tracker          # noqa: E402

POLICIES = {'sca': E.SCAPolicy, 'rvo': E.RVO3DPolicy, 'srvo': E.SRVO3DPolicy, 'orca': E.ORCA3DPolicy,
            'orca-lp': E.ORCA3DPolicyOfficial, 'rvo-dubins': E.RVO3dDubinsPolicy}


def build_obstacles():
    """run_sca.py:129-155 (exp2): eight spheres of radius 1 on a ring of radius 4 at z = 5."""
    out = []
    for j in range(8):
        out.append(E.Obstacle(pos=[round(4.0 * math.cos(2 * j * math.pi / 8), 2), round(4.0 * math.sin(2 * j * math.pi / 8), 2), 5.0],
                              shape_dict={'shape': 'sphere', 'feature': 1.0}, id=j))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenario', default='circle', choices=['circle', 'takeoff', 'exp3', 'sphere', 'random'])
    ap.add_argument('--agents', type=int, default=16)
    ap.add_argument('--policy', default='sca', choices=sorted(POLICIES))
    ap.add_argument('--no-obstacles', action='store_true')
    ap.add_argument('--map', default=None, help='binvox map for --scenario exp3')
    ap.add_argument('--max-steps', type=int, default=100000)
    ap.add_argument('--log-dir', default='visualization/sca/log')
    ap.add_argument('--tracker', default='host', choices=['host', 'device'],
                    help='where the Dubins v_pref tracker of SCA / RVO3D+Dubins runs: host = native threads, bit-exact against '
                         'the reference; device = kernels inside every step (large swarms)')
    args = ap.parse_args()

    n = args.agents
    if args.scenario == 'circle':
        sc = scenarios.circle(n, rad=10.0 if n <= 16 else None)
        obstacles = [] if args.no_obstacles else build_obstacles()
    elif args.scenario == 'takeoff':
        sc = scenarios.takeoff_landing(n)
        obstacles = [] if args.no_obstacles else [E.Obstacle(pos=list(p), shape_dict={'shape': 'sphere', 'feature': float(r)}, id=j)
                                                  for j, (p, r) in enumerate(zip(sc['obs_pos'], sc['obs_radius']))]
    elif args.scenario == 'exp3':
        sc = scenarios.spawn_n_drones(n)
        obstacles = read_map.read_obstacle(center=(35, 30), environ='exp3', obs_path=args.map or 'visualization/map/map.binvox')
    elif args.scenario == 'sphere':
        sc = scenarios.sphere(n)
        obstacles = []
    else:
        sc = scenarios.random_cube(n)
        obstacles = []
    pol = POLICIES[args.policy]
    agents = [E.Agent(start_pos=list(sc['start'][i]), goal_pos=list(sc['goal'][i]), vel=[0.0, 0.0, 0.0], radius=0.5, pref_speed=1.0,
                      policy=pol, id=i) for i in range(n)]

    v_pref_fn = None
    dubins = pol in (E.SCAPolicy, E.RVO3dDubinsPolicy)          # these two follow a Dubins path (scaPolicy.py:264-338)
    if dubins and args.tracker == 'host':
        v_pref_fn = tracker.DubinsTracker(sc['goal'][:, :3], sc['goal'][:, 3:6], np.ones(n), S.zaxis_flags(sc['start'], sc['goal']))
    env = E.MACAEnv(v_pref_fn=v_pref_fn, history_capacity=args.max_steps, device_tracker=dubins and args.tracker == 'device')
    env.set_agents(agents, obstacles=obstacles)

    step, t0 = 0, time.time()
    while step < args.max_steps:
        done = env.step({})
        step += 1
        if done:
            break
    cost = time.time() - t0
    print('All agents finished!' if done else 'step limit reached', step, 'steps,', f'{cost:.2f} s')
    paths = metrics.write_episode_log(env, args.log_dir)                  # AverageCost from agent.total_time, as run_sca.py:241-250
    info = metrics.episode_metrics(env)
    print({k: info[k] for k in ('SuccessRate', 'ExtraTime', 'ExtraDistance', 'AverageSpeed', 'AverageCost')})
    print('wrote', ', '.join(sorted(paths.values())))


if __name__ == '__main__':
    main()
