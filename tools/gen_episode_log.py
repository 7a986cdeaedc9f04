#!/usr/bin/env python3
"""Golden vectors for the episode log.  Runs ONLY in the build container.

Runs whole reference episodes with the reference's own logger active (Agent.to_vector -> history_info,
agent.py:126-148; only the removed pandas `DataFrame.append` is shimmed, as SURVEY §8(c) prescribes) and then
executes the metrics block of the reference's run script (run_example/run_sca.py `__main__`, "scenario information"
… up to the json dump) on the finished agents.  Recorded, per scenario:

  hist      [n, max_rows, 13]  the 13 ANIMATION_COLUMNS of every agent's history_info (NaN padded)
  rows      [n]                rows per agent (== env steps run: done agents keep logging)
  step_num  [n]                agent.step_num (stops counting at the goal, mampenv.py:44-45)
  env_cfg   json string        the dict the reference writes to env_cfg.json (wall-clock fields zeroed)
  start / goal6 / radius / pref_speed / policy / obs_pos / obs_radius   the scenario itself

Only data is written to tests/golden/F11_*.npz.
"""
import contextlib
import io
import json
import os
import sys
import textwrap

import numpy as np

REF = '/root/reference'
COLS = ['pos_x', 'pos_y', 'pos_z', 'alpha', 'beta', 'gamma', 'vel_x', 'vel_y', 'vel_z', 'gol_x', 'gol_y', 'gol_z', 'radius']


def main():
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, 'run_example'))
    import matplotlib
    matplotlib.use('Agg')
    import pandas as pd
    pd.DataFrame.append = lambda s, o, ignore_index=False: pd.concat([s, pd.DataFrame(o)], ignore_index=ignore_index)
    import run_sca as rs
    from mamp.envs.mampenv import MACAEnv
    from mamp.configs.config import DT

    src = open(os.path.join(REF, 'run_example', 'run_sca.py')).read().split('\n')
    b = next(i for i, l in enumerate(src) if l.strip() == '# scenario information')
    e = next(i for i, l in enumerate(src) if l.strip().startswith('info_str = json.dumps'))
    metrics_block = textwrap.dedent('\n'.join(src[b:e]))

    def run(name, agents, obstacles, max_steps):
        env = MACAEnv()
        with contextlib.redirect_stdout(io.StringIO()):
            env.set_agents(agents, obstacles=obstacles)
            for step in range(max_steps):
                if env.step({}):
                    break
        n = len(agents)
        rows = np.array([len(a.history_info) for a in agents], np.int32)
        assert all(int(r) == step + 1 for r in rows)      # update_velocitie logs every agent every step (mampenv.py:42-43)
        hist = np.full((n, int(rows.max()), len(COLS)), np.nan)
        for i, a in enumerate(agents):
            assert list(a.history_info.columns) == COLS
            hist[i, :rows[i]] = a.history_info.to_numpy(dtype=np.float64)
        ns = {'agents': agents, 'obstacles': obstacles, 'agents_num': n, 'DT': DT}
        exec(metrics_block, ns)
        cfg = ns['info_dict_to_visualize']
        cfg['all_compute_time'] = 0.0          # wall clock of the Python policy calls: not reproducible
        cfg['AverageCost'] = 0.0
        from mamp.policies.sca.scaPolicy import SCAPolicy
        np.savez_compressed(
            os.path.join('tests', 'golden', name + '.npz'), hist=hist, rows=rows, step_num=np.array([a.step_num for a in agents], np.int32),
            total_dist=np.array([a.total_dist for a in agents]), env_cfg=json.dumps(cfg),
            start=np.array([a.initial_pos for a in agents], np.float64), goal6=np.array([a.goal_pos for a in agents], np.float64),
            radius=np.array([a.radius for a in agents]), pref_speed=np.array([a.pref_speed for a in agents]),
            policy=np.zeros(n, np.uint8),
            obs_pos=np.array([o.pos_global_frame for o in obstacles], np.float64).reshape(len(obstacles), 3),
            obs_radius=np.array([o.radius for o in obstacles], np.float64), steps_run=step + 1,
            numpy_version=np.__version__)
        print(name, 'steps', step + 1, 'rows', rows.tolist(), {k: cfg[k] for k in ('SuccessRate', 'ExtraTime', 'ExtraDistance', 'AverageSpeed')})

    # F11a: the reference's default run (run_sca.py build_agents(): circle N=16 rad 10, build_obstacles(): 8 spheres)
    run('F11_episode_log_circle16_obs', rs.build_agents(), rs.build_obstacles(), 600)
    # F11b: take-off / landing with the same obstacle spheres (run_sca.py:53-81,139-150)
    from mamp.agents.agent import Agent
    from mamp.policies.sca.scaPolicy import SCAPolicy
    pos, goal = rs.set_takeoff_landing_pos(16)
    agents = [Agent(start_pos=pos[i], goal_pos=goal[i], vel=[0.0, 0.0, 0.0], radius=0.5, pref_speed=1.0, policy=SCAPolicy, id=i, dt=DT)
              for i in range(len(pos))]
    run('F11_episode_log_takeoff16', agents, rs.build_obstacles(), 600)


if __name__ == '__main__':
    main()
