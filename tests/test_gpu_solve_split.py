"""k_solve in two launches (k_solve_sweep beside the tracker's re-plans, k_solve_pick4 behind them; sca_kernels.hip.h,
solve_fast): the split must not change a bit.  SCA_SOLVE_SPLIT=1 forces it for every pass (it is normally chosen only for
large tracked shards), =0 forbids it.

* every golden episode fixture of the reference through the forced split: actions, n_suit, fallback flags as recorded from
  scaPolicy.py / rvo3dPolicy.py / srvo3dPolicy.py / orca3dPolicy.py;
* big scenes of every policy, stepped resident for whole episodes in both forms: state, actions, diagnostics equal;
* the real thing: a tracked shard with the sweep on the side stream beside the re-plans against the one-launch form.
"""
import numpy as np
import pytest

from golden_util import episode_fixtures, load, static_inputs
from test_gpu_parity import check_actions, make_solver

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def S():
    import sca_amd.solver as S
    return S


@pytest.mark.parametrize('name', episode_fixtures())
def test_forced_split_vs_golden(S, name, monkeypatch):
    monkeypatch.setenv('SCA_SOLVE_SPLIT', '1')
    fx = load(name)
    st = static_inputs(fx)
    sol = make_solver(S, fx, st)
    T = len(fx['step'])
    for t in range(0, T, max(1, T // 40)):
        sol.set_state(fx['pos'][t], fx['vel'][t], fx['heading'][t], fx['flags'][t], fx['total_dist'][t])
        sol.set_kd_perm(fx['perm'][t])
        sol.set_vpref(fx['vpref'][t], st['vpref_mode'])
        sol.policy_pass(S.NBR_KDTREE)
        ctx = (name, t)
        dg = sol.diag()
        sel = fx['n_suit'][t] >= 0
        assert np.array_equal(dg['diag'][sel, 0], fx['n_suit'][t][sel]), ctx + ('n_suit',)
        assert np.array_equal(dg['diag'][sel, 1], fx['fallback'][t][sel]), ctx + ('fallback',)
        check_actions(sol.actions(), fx['action'][t], ctx)
    sol.close()


def _pair(S, monkeypatch, sc, policy, tracker=False, params=None):
    from sca_amd import scenarios
    n = len(sc['start'])
    sols = []
    for split in ('0', '1'):
        monkeypatch.setenv('SCA_SOLVE_SPLIT', split)                      # read by sca_create
        sol = S.BatchedSolver(max_agents=n, max_obstacles=max(1, len(sc['obs_radius'])), params=params)
        sol.set_obstacles(sc['obs_pos'], sc['obs_radius'])
        sol.set_agents(np.full(n, 0.5), np.ones(n), sc['goal'][:, :3], policy, S.zaxis_flags(sc['start'], sc['goal']),
                       scenarios.max_run_dist(sc['start'], sc['goal']))
        sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
        if tracker:
            sol.device_tracker_enable(sc['goal'][:, 3:6])
        sols.append(sol)
    return sols


def _same(a, b, ctx):
    sa, sb = a.get_state(), b.get_state()
    for k in ('pos', 'vel', 'heading', 'flags', 'total_dist', 'step_num'):
        assert np.array_equal(sa[k], sb[k]), ctx + (k,)
    assert np.array_equal(a.actions(), b.actions(), equal_nan=True), ctx + ('actions',)
    da, db = a.diag(), b.diag()
    assert np.array_equal(da['diag'], db['diag']), ctx + ('diag',)
    assert np.array_equal(da['status'], db['status']), ctx + ('status',)
    assert np.array_equal(da['vpref'], db['vpref'], equal_nan=True), ctx + ('vpref',)


@pytest.mark.parametrize('kind,n,pol,steps', [('circle', 2048, 0, 40), ('random', 4096, 1, 25), ('random', 4096, 2, 25),
                                              ('random', 4096, 3, 25), ('takeoff', 1024, -1, 60), ('random', 20000, -2, 12)])
def test_split_episode_equals_one_launch_episode(S, monkeypatch, kind, n, pol, steps):
    from sca_amd import scenarios
    sc = {'circle': lambda: scenarios.circle(n), 'random': lambda: scenarios.random_cube(n, seed=5),
          'takeoff': lambda: scenarios.takeoff_landing(n)}[kind]()
    nn = len(sc['start'])
    if pol == -1:
        policy = np.where(np.arange(nn) % 2 == 0, 0, 2)
    elif pol == -2:
        policy = np.arange(nn) % 5                                      # all five policies in one shard, LP agents included
    else:
        policy = np.full(nn, pol)
    one, two = _pair(S, monkeypatch, sc, policy.astype(np.uint8))
    for t in range(steps):
        one.run_steps(1)
        two.run_steps(1)
        one.synchronize(); two.synchronize()
        _same(one, two, (kind, n, pol, t))
    assert one.active_count() == two.active_count()
    one.close(); two.close()


@pytest.mark.parametrize('pol', [0, 1, 2, 3])
def test_split_with_the_heading_limit_off(S, monkeypatch, pol):
    """max_heading_change = pi: the posture constraint of util.py:6-20 passes every direction, the survivor lists grow to
    hundreds of candidates -- beyond what k_solve_pick4 caches per agent (PICK_CAP), the recomputing path."""
    import math
    from sca_amd import scenarios
    sc = scenarios.random_cube(1500, seed=11)
    one, two = _pair(S, monkeypatch, sc, np.full(1500, pol, np.uint8), params=dict(max_heading_change=math.pi))
    longest = 0
    for t in range(12):
        one.run_steps(1)
        two.run_steps(1)
        one.synchronize(); two.synchronize()
        _same(one, two, ('wide', pol, t))
        longest = max(longest, int(one.diag()['diag'][:, 0].max()))
    assert longest > 300, longest
    one.close(); two.close()


@pytest.mark.parametrize('n', [700, 30000])
def test_split_beside_the_replans_equals_one_launch(S, monkeypatch, n):
    """Tracker in the pass: k_solve_sweep runs on the neighbour stream while the re-plans run on the main one
    (30 000 agents: the lane-per-plan kernel, the configuration the split is chosen for; 700: the speculative forms)."""
    from sca_amd import scenarios
    sc = scenarios.circle(n)
    policy = np.where(np.arange(n) % 7 == 3, 3, 0).astype(np.uint8)       # SCA, every seventh agent ORCA3D (not tracked)
    one, two = _pair(S, monkeypatch, sc, policy, tracker=True)
    for t in range(14):
        one.run_steps(1)
        two.run_steps(1)
        one.synchronize(); two.synchronize()
        _same(one, two, ('tracked', n, t))
    assert np.array_equal(one.device_tracker_replans(), two.device_tracker_replans())
    one.close(); two.close()


def test_forms_the_library_picks_by_itself(S):
    """No switches: a large tracked shard on the circle (nearly every agent re-plans every step) gets the lane-per-plan re-plan
    kernel behind k_track (the fused k_track_replan is opt-in since round 3: SCA_TRACKER_FUSE) and the split solve once the
    first re-plan count has come back; a small one keeps k_track + k_replan_few and the one-launch k_solve; without the tracker
    nothing is split."""
    from sca_amd import scenarios

    def run(n, tracker, steps):
        sc = scenarios.circle(n)
        sol = S.BatchedSolver(max_agents=n)
        sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
        sol.set_agents(np.full(n, 0.5), np.ones(n), sc['goal'][:, :3], np.zeros(n, np.uint8), S.zaxis_flags(sc['start'], sc['goal']),
                       scenarios.max_run_dist(sc['start'], sc['goal']))
        sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
        if tracker:
            sol.device_tracker_enable(sc['goal'][:, 3:6])
        for _ in range(steps):
            sol.run_steps(1)
            sol.synchronize()
        f = sol.pass_forms()
        sol.close()
        return f

    big = run(60000, True, 16)
    assert big & S.FORM_SOLVE_SPLIT and big & S.FORM_REPLAN_LANE and not big & (S.FORM_REPLAN_FEW | S.FORM_TRACK_FUSED), big
    small = run(3000, True, 16)
    assert small & S.FORM_REPLAN_FEW and not small & (S.FORM_SOLVE_SPLIT | S.FORM_TRACK_FUSED | S.FORM_REPLAN_LANE), small
    assert run(60000, False, 3) == 0


def test_long_tracked_episode_all_forms_against_the_plain_ones(S, monkeypatch):
    """A soak for the stream choreography: 240 resident steps of a 60 000-agent tracked shard with everything the library
    can pick (split solve on two streams, k_track_replan switched on, count readbacks, form switches as the re-plan count moves)
    against the same episode with the plain forms (k_track + k_replan, one-launch k_solve), compared every 20 steps."""
    from sca_amd import scenarios
    n = 60000
    sc = scenarios.circle(n)
    policy = np.where(np.arange(n) % 13 == 6, 3, 0).astype(np.uint8)
    sols = []
    for plain in (True, False):
        if plain:
            monkeypatch.setenv('SCA_SOLVE_SPLIT', '0'); monkeypatch.setenv('SCA_TRACKER_NOFUSE', '1'); monkeypatch.delenv('SCA_TRACKER_FUSE', raising=False)
        else:                                          # (the fused tracker kernel is opt-in since round 3: covered here)
            monkeypatch.delenv('SCA_SOLVE_SPLIT', raising=False); monkeypatch.delenv('SCA_TRACKER_NOFUSE', raising=False); monkeypatch.setenv('SCA_TRACKER_FUSE', '1')
        sol = S.BatchedSolver(max_agents=n)
        sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
        sol.set_agents(np.full(n, 0.5), np.ones(n), sc['goal'][:, :3], policy, S.zaxis_flags(sc['start'], sc['goal']),
                       scenarios.max_run_dist(sc['start'], sc['goal']))
        sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
        sol.device_tracker_enable(sc['goal'][:, 3:6])
        sols.append(sol)
    a, b = sols
    seen = 0
    for blk in range(12):
        a.run_steps(20); b.run_steps(20)
        a.synchronize(); b.synchronize()
        _same(a, b, ('soak', blk))
        seen |= b.pass_forms()
    assert seen & S.FORM_SOLVE_SPLIT and seen & S.FORM_TRACK_FUSED, seen
    assert np.array_equal(a.device_tracker_replans(), b.device_tracker_replans())
    a.close(); b.close()
