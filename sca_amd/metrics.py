"""Episode log of the reference's run scripts, produced from device state.

  * `episode_metrics(env)`  — SuccessRate / ExtraTime / ExtraDistance / AverageSpeed / AverageCost, the formulas of
    run_example/run_sca.py:223-251 (identical blocks in run_rvo.py, run_srvo.py, run_orca.py, run_rvodubins.py).
  * `episode_info(env)`     — the dict the reference dumps to `env_cfg.json` (run_sca.py:199-259), same keys and order, which
    visualization/draw_episode.py:17-32 reads (`all_agent_info`, `all_obstacle`).
  * `trajectories(env)`     — `agent.history_info` (agent.py:75-77,126-148): the 13 ANIMATION_COLUMNS per agent per env step,
    read back from the log the integrate kernel keeps in HBM (`sca_history_enable`), so a resident run needs no per-step
    readback.
  * `write_episode_log(env, dir)` — `env_cfg.json` + `trajs.npz` (one [rows, 13] array per agent under the reference's
    sheet name `agent<id>`); `trajs.xlsx` as well when openpyxl is importable (it is what the reference writes).

AverageCost is the reference's wall time of find_next_action per agent-step (run_sca.py:250): by default the sum of
`agent.total_time`, which MACAEnv.step maintains (a step's policy wall time shared among the agents it served); a measured
total can be passed instead.
"""
import json
import os

import numpy as np

from .env import DT

ANIMATION_COLUMNS = ['pos_x', 'pos_y', 'pos_z', 'alpha', 'beta', 'gamma', 'vel_x', 'vel_y', 'vel_z',
                     'gol_x', 'gol_y', 'gol_z', 'radius']                                     # agent.py:75-76


def episode_metrics(env, total_policy_time_s=None):
    agents = env.agents
    n = len(agents)
    ok = np.array([(not a.is_collision) and (not a.is_out_of_max_time) for a in agents])
    num = int(ok.sum())
    # the reference accumulates in agent order with Python floats (run_sca.py:232-240): keep the same summation order
    straight = 0.0
    dist = 0.0
    desire = 0
    steps = 0
    for a, k in zip(agents, ok):
        if k:
            straight += a.straight_path_length
            dist += float(a.total_dist)
            desire += a.desire_steps
            steps += int(a.step_num)
    out = {
        'successful_num': num, 'all_straight_distance': straight, 'all_distance': dist, 'all_desire_step_num': desire,
        'all_step_num': steps, 'SuccessRate': num / n,
        'ExtraTime': ((steps - desire) * DT) / num if num else float('nan'),
        'ExtraDistance': (dist - straight) / num if num else float('nan'),
        'AverageSpeed': dist / steps / DT if steps else float('nan'),
    }
    if total_policy_time_s is None:                      # run_sca.py:241-250: sum of agent.total_time over the successful agents
        total_policy_time_s = sum(a.total_time for a, k in zip(agents, ok) if k)
    if steps:
        out['AverageCost'] = 1000 * total_policy_time_s / steps
    return out


def episode_info(env, total_policy_time_s=None):
    """The `info_dict_to_visualize` of run_sca.py:199-259."""
    if total_policy_time_s is None:                      # run_sca.py:241: all_compute_time sums agent.total_time
        ok = [(not a.is_collision) and (not a.is_out_of_max_time) for a in env.agents]
        total_policy_time_s = sum(a.total_time for a, k in zip(env.agents, ok) if k)
    m = episode_metrics(env, total_policy_time_s)
    info = {
        'all_agent_info': [{'id': a.id, 'gp': a.group, 'radius': a.radius, 'goal_pos': np.asarray(a.goal_global_frame).tolist()}
                           for a in env.agents],
        'all_obstacle': [],
        'all_compute_time': float(total_policy_time_s),
        'all_straight_distance': m['all_straight_distance'],
        'all_distance': m['all_distance'],
        'successful_num': m['successful_num'],
        'all_desire_step_num': m['all_desire_step_num'],
        'all_step_num': m['all_step_num'],
        'SuccessRate': m['SuccessRate'],
        'ExtraTime': m['ExtraTime'],
        'ExtraDistance': m['ExtraDistance'],
        'AverageSpeed': m['AverageSpeed'],
        'AverageCost': m.get('AverageCost', 0.0),
    }
    for o in env.obstacles:
        info['all_obstacle'].append({'position': list(o.pos), 'shape': o.shape, 'feature': o.feature})
    return info


def trajectories(env, agent_begin=0, agent_count=None):
    """[agents, rows, 13] array of the ANIMATION_COLUMNS, read from the device log (needs MACAEnv(history_capacity=...))."""
    rows, dropped = env.solver.history_rows()
    if dropped:
        raise RuntimeError(f'{dropped} env steps did not fit the trajectory log: raise history_capacity')
    n = len(env.agents)
    if agent_count is None:
        agent_count = n - agent_begin
    h = env.solver.history(0, rows, agent_begin, agent_count)
    out = np.empty((agent_count, rows, len(ANIMATION_COLUMNS)))
    out[:, :, 0:3] = h['pos'].transpose(1, 0, 2)
    out[:, :, 3:6] = h['heading'].transpose(1, 0, 2)
    out[:, :, 6:9] = h['vel'].transpose(1, 0, 2)            # float32 values, as agent.vel_global_frame holds them
    out[:, :, 9:12] = env.goal[agent_begin:agent_begin + agent_count, None, :]
    out[:, :, 12] = np.array([a.radius for a in env.agents[agent_begin:agent_begin + agent_count]])[:, None]
    return out


def write_episode_log(env, log_dir, total_policy_time_s=None, xlsx=None):
    """Writes what run_sca.py:181-259 writes: the trajectories and env_cfg.json.  Returns the paths."""
    os.makedirs(log_dir, exist_ok=True)
    paths = {}
    traj = trajectories(env)
    paths['trajs'] = os.path.join(log_dir, 'trajs.npz')
    np.savez_compressed(paths['trajs'], columns=np.array(ANIMATION_COLUMNS),
                        **{'agent' + str(a.id): traj[i] for i, a in enumerate(env.agents)})
    if xlsx is None or xlsx:
        try:
            import openpyxl  # noqa: F401
            import pandas as pd
            paths['xlsx'] = os.path.join(log_dir, 'trajs.xlsx')
            with pd.ExcelWriter(paths['xlsx']) as writer:
                for i, a in enumerate(env.agents):
                    pd.DataFrame(traj[i], columns=ANIMATION_COLUMNS).to_excel(writer, sheet_name='agent' + str(a.id))
        except ImportError:
            if xlsx:
                raise
    paths['env_cfg'] = os.path.join(log_dir, 'env_cfg.json')
    with open(paths['env_cfg'], 'w') as f:
        f.write(json.dumps(episode_info(env, total_policy_time_s), indent=4))
    return paths


def read_trajs(path):
    """trajs.npz -> what draw_episode.get_agent_traj builds from the xlsx: list of {column: list} per agent, in file order."""
    z = np.load(path)
    cols = [str(c) for c in z['columns']]
    keys = sorted((k for k in z.files if k.startswith('agent')), key=lambda k: int(k[5:]))
    return [{c: z[k][:, j].tolist() for j, c in enumerate(cols)} for k in keys]
