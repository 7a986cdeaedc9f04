"""Episode metrics of the reference's run scripts (run_example/run_sca.py:199-259), computed from the env's arrays.

SuccessRate / ExtraTime / ExtraDistance / AverageSpeed follow the reference's formulas literally; AverageCost (the
reference's wall time of find_next_action per agent-step, run_sca.py:250) is reported from a caller-supplied total time.
"""
import numpy as np

from .env import DT


def episode_metrics(env, total_policy_time_s=None):
    agents = env.agents
    n = len(agents)
    ok = np.array([(not a.is_collision) and (not a.is_out_of_max_time) for a in agents])
    num = int(ok.sum())
    straight = sum(a.straight_path_length for a, k in zip(agents, ok) if k)
    dist = float(env.total_dist[ok].sum())
    desire = sum(a.desire_steps for a, k in zip(agents, ok) if k)
    steps = int(env.step_num[ok].sum())
    out = {
        'successful_num': num, 'all_straight_distance': straight, 'all_distance': dist, 'all_desire_step_num': desire,
        'all_step_num': steps, 'SuccessRate': num / n,
        'ExtraTime': ((steps - desire) * DT) / num if num else float('nan'),
        'ExtraDistance': (dist - straight) / num if num else float('nan'),
        'AverageSpeed': dist / steps / DT if steps else float('nan'),
    }
    if total_policy_time_s is not None and steps:
        out['AverageCost'] = 1000 * total_policy_time_s / steps
    return out
