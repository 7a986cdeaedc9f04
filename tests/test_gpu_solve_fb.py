"""k_solve_fb (round 4): shards small enough for every wavefront to be resident at once solve AND finish their fallbacks (agents
without any suitable candidate: the complete sweep incl. compute_without_suitV, scaPolicy.py:224-238) in one launch instead of
k_solve -> list -> k_fallback.  SCA_SOLVE_FB_MAX=0 forbids it: both forms must leave the same bits.
k_action_fb (round 6): shards of up to 16 384 agents that do go through the list run the fallback sweep inside the epilogue's launch
instead of in a launch of its own in front of it; SCA_ACTION_FB_MAX=0 forbids that.  Three forms, the same bits.

* every golden episode fixture both ways: n_suit, fallback flags and action rows as recorded from the reference's policies;
* a dense scene (many fallbacks) of SCA, RVO3D and S-RVO3D agents stepped resident in both forms."""
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
def test_both_forms_vs_golden(S, name, monkeypatch):
    fx = load(name)
    st = static_inputs(fx)
    has_lp = bool(np.any(np.asarray(st['policy']) == S.POL_ORCA3D_LP)) if 'policy' in st else False
    for cap, acap, want in (('0', '0', False), ('0', '1000000', False), ('1000000', '1000000', True)):
        monkeypatch.setenv('SCA_SOLVE_FB_MAX', cap)                      # read by sca_create
        monkeypatch.setenv('SCA_ACTION_FB_MAX', acap)
        sol = make_solver(S, fx, st)
        T = len(fx['step'])
        fused_seen = False
        for t in range(0, T, max(1, T // 25)):
            sol.set_state(fx['pos'][t], fx['vel'][t], fx['heading'][t], fx['flags'][t], fx['total_dist'][t])
            sol.set_kd_perm(fx['perm'][t])
            sol.set_vpref(fx['vpref'][t], st['vpref_mode'])
            sol.policy_pass(S.NBR_KDTREE)
            fused_seen |= bool(sol.pass_forms() & S.FORM_SOLVE_FB)
            assert bool(sol.pass_forms() & S.FORM_ACTION_FB) == (acap != '0' and not (sol.pass_forms() & S.FORM_SOLVE_FB)), (name, t, cap, acap)
            ctx = (name, t, cap, acap)
            dg = sol.diag()
            sel = fx['n_suit'][t] >= 0
            assert np.array_equal(dg['diag'][sel, 0], fx['n_suit'][t][sel]), ctx + ('n_suit',)
            assert np.array_equal(dg['diag'][sel, 1], fx['fallback'][t][sel]), ctx + ('fallback',)
            check_actions(sol.actions(), fx['action'][t], ctx)
        assert fused_seen == (want and not has_lp), (name, cap, fused_seen)
        sol.close()


def test_dense_scene_with_fallbacks_both_forms(S, monkeypatch):
    from sca_amd import scenarios
    n = 900
    rng = np.random.default_rng(12)
    start = np.zeros((n, 6)); goal = np.zeros((n, 6))
    start[:, :3] = rng.uniform(-9, 9, (n, 3)) + [0, 0, 20]                # ~60 agents within neighborDist: most sweeps end empty-handed
    goal[:, :3] = -start[:, :3] + [0, 0, 40]
    start[:, 3] = rng.uniform(0, 2 * np.pi, n)
    policy = (np.arange(n) % 3).astype(np.uint8)                          # SCA (straight v_pref), RVO3D, S-RVO3D
    sols = []
    for cap, acap in (('0', '0'), ('1000000', '0'), ('0', '1000000')):
        monkeypatch.setenv('SCA_SOLVE_FB_MAX', cap)
        monkeypatch.setenv('SCA_ACTION_FB_MAX', acap)
        sol = S.BatchedSolver(max_agents=n, max_obstacles=1)
        sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
        sol.set_agents(np.full(n, 0.5), np.ones(n), goal[:, :3], policy, S.zaxis_flags(start, goal), scenarios.max_run_dist(start, goal))
        sol.set_state(start[:, :3], np.zeros((n, 3), np.float32), start[:, 3:6], np.zeros(n, np.uint8))
        sols.append(sol)
    fallbacks = 0
    for t in range(40):
        for sol in sols:
            sol.run_steps(1)
            sol.synchronize()
        a, da = sols[0].get_state(), sols[0].diag()
        for other in sols[1:]:
            b, db = other.get_state(), other.diag()
            for k in ('pos', 'vel', 'heading', 'flags', 'total_dist', 'step_num'):
                assert np.array_equal(a[k], b[k]), (t, k)
            assert np.array_equal(da['diag'], db['diag']), t
        fallbacks += int((da['diag'][:, 1] == 1).sum())
        f = [sol.pass_forms() for sol in sols]           # k_fallback + k_action | k_solve_fb | k_action_fb
        assert not (f[0] & (S.FORM_SOLVE_FB | S.FORM_ACTION_FB)) and (f[1] & S.FORM_SOLVE_FB) and not (f[1] & S.FORM_ACTION_FB)
        assert (f[2] & S.FORM_ACTION_FB) and not (f[2] & S.FORM_SOLVE_FB)
    assert fallbacks > 200, fallbacks                                     # the scene is there for them
    for sol in sols:
        sol.close()
