"""Episode log: metrics dict / env_cfg.json schema / trajectory log against what the reference itself
produced (tools/gen_episode_log.py ran whole reference episodes with its logger on and then executed the metrics block of
run_example/run_sca.py; fixtures tests/golden/F11_episode_log_*.npz).

CPU part: the host-side formulas on the reference's final agent state.  GPU part: the whole episode on the device
(SCAPolicy + native tracker), trajectories read back from the HBM log."""
import json
import os
import types

import numpy as np
import pytest

from golden_util import GOLDEN

FIXTURES = ['F11_episode_log_circle16_obs', 'F11_episode_log_takeoff16']


def _load(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return {k: z[k] for k in z.files}


def _agents_and_obstacles(fx, E):
    n = len(fx['rows'])
    agents = [E.Agent(start_pos=list(fx['start'][i]), goal_pos=list(fx['goal6'][i]), vel=[0.0, 0.0, 0.0], radius=float(fx['radius'][i]),
                      pref_speed=float(fx['pref_speed'][i]), policy=E.SCAPolicy, id=i) for i in range(n)]
    obstacles = [E.Obstacle(pos=[float(v) for v in fx['obs_pos'][j]], shape_dict={'shape': 'sphere', 'feature': float(fx['obs_radius'][j])}, id=j)
                 for j in range(len(fx['obs_radius']))]
    return agents, obstacles


@pytest.mark.parametrize('name', FIXTURES)
def test_episode_info_matches_reference_env_cfg(name):
    """Same dict, same key order, same floats as the reference's env_cfg.json, from the reference's final agent state."""
    from sca_amd import env as E, metrics
    fx = _load(name)
    agents, obstacles = _agents_and_obstacles(fx, E)
    for i, a in enumerate(agents):
        a._total_dist = float(fx['total_dist'][i])
        a._step_num = int(fx['step_num'][i])
    env = types.SimpleNamespace(agents=agents, obstacles=obstacles)
    want = json.loads(str(fx['env_cfg']))
    got = json.loads(json.dumps(metrics.episode_info(env)))
    assert list(got.keys()) == list(want.keys())
    for k in want:
        assert got[k] == want[k], k
    m = metrics.episode_metrics(env, total_policy_time_s=2.0)
    assert m['AverageCost'] == 1000 * 2.0 / want['all_step_num']


def test_metrics_with_failures():
    """Collided / timed-out agents are left out of every sum (run_sca.py:229-236); SuccessRate counts them."""
    from sca_amd import env as E, metrics
    fx = _load(FIXTURES[0])
    agents, obstacles = _agents_and_obstacles(fx, E)
    for i, a in enumerate(agents):
        a._total_dist = float(fx['total_dist'][i])
        a._step_num = int(fx['step_num'][i])
    agents[3]._flags = 2          # collision
    agents[5]._flags = 4          # out of max time
    env = types.SimpleNamespace(agents=agents, obstacles=obstacles)
    m = metrics.episode_metrics(env)
    ok = [i for i in range(len(agents)) if i not in (3, 5)]
    assert m['successful_num'] == len(ok) and m['SuccessRate'] == len(ok) / len(agents)
    assert m['all_step_num'] == int(fx['step_num'][ok].sum())
    assert m['all_distance'] == sum(float(fx['total_dist'][i]) for i in ok)


@pytest.mark.gpu
@pytest.mark.parametrize('name', FIXTURES)
def test_device_trajectory_log_follows_reference_history(name, tmp_path):
    """Whole SCA episode on the device with the HBM trajectory log on; the log must be the reference's history_info, the
    written env_cfg.json the reference's, and the files must read back the way visualization/draw_episode.py:17-32 consumes
    them.  Round 6: EQUALITY over the whole free-running episode -- positions, headings, float32 velocities, flags of every step (until
    round 5 the device's sin / cos / atan2 left positions 1e-14 away from the host libm's, and at step 151 of the circle run one v_pref
    component sat within that distance of a 5-decimal truncation edge: tolerances of 1e-5 … 5e-5 covered it)."""
    from sca_amd import env as E, metrics, tracker, solver as S
    fx = _load(name)
    agents, obstacles = _agents_and_obstacles(fx, E)
    n = len(agents)
    goal = fx['goal6'][:, :3]
    tr = tracker.DubinsTracker(goal, fx['goal6'][:, 3:6], fx['pref_speed'], S.zaxis_flags(fx['start'], fx['goal6']), nthreads=1)
    env = E.MACAEnv(v_pref_fn=tr, history_capacity=700)
    env.set_agents(agents, obstacles=obstacles)
    steps = 0
    while steps < 700:
        steps += 1
        if env.step({}):
            break
    assert steps == int(fx['steps_run'])
    traj = metrics.trajectories(env)
    want = fx['hist']
    assert traj.shape == want.shape
    for lo, hi, what in ((0, 3, 'pos'), (3, 6, 'heading'), (6, 9, 'vel')):
        assert np.array_equal(traj[:, :, lo:hi], want[:, :, lo:hi]), (name, what, float(np.abs(traj[:, :, lo:hi] - want[:, :, lo:hi]).max()))
    assert np.array_equal(traj[:, :, 9:13], want[:, :, 9:13])
    assert np.array_equal(env.step_num, fx['step_num'])
    # log == the state the env reports after each step (last row = final state)
    assert np.array_equal(traj[:, -1, 0:3], env.pos) and np.array_equal(traj[:, -1, 6:9], env.vel.astype(np.float64))
    paths = metrics.write_episode_log(env, str(tmp_path), xlsx=False)
    cfg = json.load(open(paths['env_cfg']))
    ref = json.loads(str(fx['env_cfg']))
    assert list(cfg.keys()) == list(ref.keys())
    for k in ('all_agent_info', 'all_obstacle', 'successful_num', 'all_desire_step_num', 'all_step_num', 'SuccessRate', 'ExtraTime',
              'all_straight_distance'):
        assert cfg[k] == ref[k], k
    for k in ('all_distance', 'ExtraDistance', 'AverageSpeed'):
        assert abs(cfg[k] - ref[k]) <= 1e-4, k
    back = metrics.read_trajs(paths['trajs'])
    assert len(back) == n and list(back[0].keys()) == metrics.ANIMATION_COLUMNS and len(back[0]['pos_x']) == steps


@pytest.mark.gpu
def test_history_resident_run_and_windows():
    """sca_run_steps (no per-step readback) fills the same log as stepping through the host API; windows, capacity overflow
    and re-enable behave as include/sca_hip.h says."""
    from sca_amd import scenarios, solver as S
    sc = scenarios.circle(300)
    n = 300

    def make():
        s = S.BatchedSolver(max_agents=n, max_obstacles=1)
        s.set_obstacles(np.zeros((0, 3)), np.zeros(0))
        s.set_agents(np.full(n, 0.5), np.ones(n), sc['goal'][:, :3], np.full(n, 1, np.uint8), np.zeros(n, np.uint8), scenarios.max_run_dist(sc['start'], sc['goal']))
        s.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
        return s
    a = make()
    a.history_enable(5)
    a.run_steps(8)
    a.synchronize()
    assert a.history_rows() == (5, 3)
    ha = a.history()
    b = make()
    b.history_enable(8)
    rows = []
    for t in range(8):
        b.policy_pass()
        b.env_update()
        st = b.get_state()
        rows.append((st['pos'].copy(), st['heading'].copy(), st['vel'].copy()))
    hb = b.history()
    assert b.history_rows() == (8, 0)
    for t in range(8):
        assert np.array_equal(hb['pos'][t], rows[t][0]) and np.array_equal(hb['heading'][t], rows[t][1]) and np.array_equal(hb['vel'][t], rows[t][2])
    for k in ('pos', 'heading', 'vel'):
        assert np.array_equal(ha[k], hb[k][:5])
    w = b.history(first_row=2, nrows=3, agent_begin=17, agent_count=40)
    assert np.array_equal(w['pos'], hb['pos'][2:5, 17:57]) and np.array_equal(w['vel'], hb['vel'][2:5, 17:57])
    with pytest.raises(RuntimeError):
        b.history(first_row=6, nrows=3)
    b.history_enable(0)
    with pytest.raises(RuntimeError):
        b.history()
