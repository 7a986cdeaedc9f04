"""Parity away from the reference's default parameters (-m gpu).  The reference keeps its solver parameters on every Agent
(mamp/agents/agent.py:24-41) and reads them per call (scaPolicy.py:95,112,272,299-302, util.py:8,17, orca3dPolicyOfficial.py:44,98,108,
agent.py:87-99, mampenv.py:90-92); the boundary exports them as sca_params + sca_device_tracker_enable's turning radius / pitch limits.
The F16 fixtures are the reference stepped with those attributes changed after Agent.__init__ (tools/gen_golden.py::_apply_attrs); they
ride in every episode-parametrised test (tests/golden_util.py::episode_fixtures) -- here: the drop-in MACAEnv built from Agent objects
that carry the changed attributes, closed loop, nothing from the fixture but the start state."""
import numpy as np
import pytest

from golden_util import fixture_agent_params, fixture_params, fixture_tracker_agent_params, hetero_fixtures, load, param_fixtures, static_inputs

pytestmark = pytest.mark.gpu

VEL_TOL = 1e-5
EPISODES = [n for n in param_fixtures() + hetero_fixtures() if len(load(n)['step']) > 1]


def _agents_from_fixture(E, fx, with_attrs=True):
    pol_cls = {0: E.SCAPolicy, 1: E.RVO3DPolicy, 2: E.SRVO3DPolicy, 3: E.ORCA3DPolicy, 4: E.ORCA3DPolicyOfficial, 5: E.RVO3dDubinsPolicy}
    n = len(fx['radius'])
    agents = [E.Agent(start_pos=list(fx['start'][i]), goal_pos=list(fx['goal6'][i]), vel=[0.0, 0.0, 0.0], radius=float(fx['radius'][i]),
                      pref_speed=float(fx['pref_speed'][i]), policy=pol_cls[int(fx['policy'][i])], id=i) for i in range(n)]
    if with_attrs:
        for i, a in enumerate(agents):                 # what a user of the reference writes: attributes set after the constructor
            for k in ('maxNeighbors', 'neighborDist', 'timeStep', 'timeHorizon', 'maxSpeed', 'min_heading_change', 'max_heading_change',
                      'turning_radius', 'dt_nominal'):
                v = fx['attr_' + k][i]
                setattr(a, k, int(v) if k == 'maxNeighbors' else float(v))
            a.pitchlims = [float(fx['attr_pitch_lo'][i]), float(fx['attr_pitch_hi'][i])]
    obstacles = [E.Obstacle(pos=list(fx['obs_pos'][j]), shape_dict={'shape': 'sphere', 'feature': float(fx['obs_radius'][j])}, id=j)
                 for j in range(len(fx['obs_radius']))]
    return agents, obstacles


@pytest.mark.parametrize('mode', ['kd', 'auto'])
@pytest.mark.parametrize('name', EPISODES)
def test_env_closed_loop_with_agent_attributes(name, mode):
    """`while not env.step()` on Agent objects with non-default attributes, the Dubins tracker on the device with the agents' turning
    radius and pitch limits: every step's velocities and positions EQUAL to the reference's (round 6: the whole
    loop runs on the restated glibc), the same agents arrive / collide / time out at the same steps."""
    from sca_amd import env as E, solver as S
    fx = load(name)
    agents, obstacles = _agents_from_fixture(E, fx)
    env = E.MACAEnv(device_tracker=True, neighbor_mode=S.NBR_AUTO if mode == 'auto' else S.NBR_KDTREE)
    env.set_agents(agents, obstacles=obstacles)
    params, trk = fixture_params(fx)
    for k, v in params.items():
        assert getattr(env.solver.params, k) == v, k
    trk_pa = ['turning_radius / pitchlims'] if fixture_tracker_agent_params(fx) else []
    assert env.per_agent_attributes == sorted(list(fixture_agent_params(fx)) + trk_pa)   # F17 / F18: what the agents disagree on went over as arrays
    T = len(fx['step'])
    assert np.array_equal(fx['step'], np.arange(T))
    worst = 0.0
    for t in range(T):
        env.step({})
        worst = max(worst, float(np.abs(env.vel - fx['vel_after'][t]).max()))
        assert worst == 0.0, (name, mode, t, worst)
        assert np.array_equal(env.flags, fx['flags_after'][t]), (name, mode, t)
        assert np.array_equal(env.pos, fx['pos_after'][t]), (name, mode, t)
    assert np.array_equal(np.array(env.kdTree.agentIDs), fx['perm_after'][-1])


def test_defaults_would_not_reproduce_these_scenes():
    """The fixtures bite: the same scenes under the reference's DEFAULT attributes leave the recorded trajectories (otherwise the tests
    above would pass with the parameters ignored)."""
    from sca_amd import env as E
    differ = 0
    for name in EPISODES:
        fx = load(name)
        agents, obstacles = _agents_from_fixture(E, fx, with_attrs=False)
        env = E.MACAEnv(device_tracker=True)
        env.set_agents(agents, obstacles=obstacles)
        for t in range(len(fx['step'])):
            env.step({})
            if np.abs(env.vel - fx['vel_after'][t]).max() > VEL_TOL or not np.array_equal(env.flags, fx['flags_after'][t]):
                differ += 1
                break
    assert differ == len(EPISODES), (differ, len(EPISODES))


def test_heterogeneous_attributes_travel_per_agent():
    """The reference reads every attribute per agent.  The solver attributes travel per agent (sca_set_agent_params, F17 fixtures), the planner's
    two as classes of equal values (sca_device_tracker_set_agent_params, F18) -- up to 16 classes; beyond, per agent (next test)."""
    from sca_amd import env as E, solver as S
    agents = E.build_circle_agents(8, policy=E.RVO3DPolicy, rad=10.0)
    agents[3].neighborDist = 5.0
    agents[5].pitchlims = [-0.3, 0.3]                              # nobody is tracked: the planner's limits are never read
    env = E.MACAEnv()
    env.set_agents(agents, obstacles=[])
    assert env.per_agent_attributes == ['neighbor_dist']
    env.step({})
    sca = E.build_circle_agents(8, policy=E.SCAPolicy, rad=10.0)
    sca[5].pitchlims = [-0.3, 0.3]
    sca[2].turning_radius = 3.0
    env = E.MACAEnv(device_tracker=True)
    env.set_agents(sca, obstacles=[])                              # three classes of (turning_radius, pitchlims) on the device
    assert env.per_agent_attributes == ['turning_radius / pitchlims']
    for _ in range(5):
        env.step({})
    # more than 16 different planner settings among the tracked agents (the reference has no limit, agent.py:24-29): round 5 refused them;
    # now the per-agent form takes over (a wavefront per plan reading its agent's own three values) -- see the test below


@pytest.mark.parametrize('n,steps', [(20, 60), (300, 12)])
def test_more_than_16_planner_settings_equal_the_host_tracker(n, steps):
    """every tracked agent its own (turning_radius, pitchlims) -- 20 / 300 distinct settings.  The device tracker's
    per-agent form against the native host tracker, which plans every agent with its own values and which the F18 fixtures pin to the
    reference: positions, velocities, flags and re-plan counts equal after every step.  (n = 300: beyond k_track_group's range, so the
    list + k_replan_group<64> launches run.)"""
    from sca_amd import env as E, scenarios, solver as S, tracker
    rng = np.random.default_rng(n)
    R = 0.8 + 2.2 * rng.random(n)
    lo = -(0.35 + 0.4 * rng.random(n))
    hi = 0.35 + 0.5 * rng.random(n)

    def run(device):
        sc = scenarios.circle(n, rad=max(10.0, 0.4 * n))
        pol = [E.SCAPolicy if i % 5 else E.RVO3dDubinsPolicy for i in range(n)]
        agents = [E.Agent(start_pos=list(sc['start'][i]), goal_pos=list(sc['goal'][i]), vel=[0.0, 0.0, 0.0], radius=0.5,
                          pref_speed=1.0, policy=pol[i], id=i) for i in range(n)]
        for i, a in enumerate(agents):
            a.turning_radius = float(R[i])
            a.pitchlims = [float(lo[i]), float(hi[i])]
        fn = None
        if not device:
            fn = tracker.DubinsTracker(sc['goal'][:, :3], sc['goal'][:, 3:6], np.ones(n), S.zaxis_flags(sc['start'], sc['goal']))
            fn.set_agent_params(R, lo, hi)
        env = E.MACAEnv(v_pref_fn=fn, device_tracker=device)
        env.set_agents(agents, obstacles=[])
        if device:
            assert env.per_agent_attributes == ['turning_radius / pitchlims']
        trail = []
        for _ in range(steps):
            env.step({})
            trail.append((env.pos.copy(), env.vel.copy(), env.flags.copy()))
        plans = env.solver.device_tracker_replans() if device else fn.replans()
        return trail, plans

    host, plans_h = run(False)
    dev, plans_d = run(True)
    for t, (h, d) in enumerate(zip(host, dev)):
        for k in range(3):
            assert np.array_equal(h[k], d[k]), (n, t, ('pos', 'vel', 'flags')[k])
    assert np.array_equal(plans_h, plans_d) and plans_h.sum() >= n


@pytest.mark.parametrize('name', EPISODES)
def test_policy_pass_grid_mode_with_parameters(name):
    """SCA_NBR_GRID sizes its cells with neighbor_dist and keeps max_neighbors entries: whenever nobody overflows (the status bit says so)
    the lists are the reference's as sets with the same distances, and the sampled policies' velocities are the reference's."""
    from sca_amd import solver as S
    from test_gpu_parity import make_solver
    fx = load(name)
    st = static_inputs(fx)
    params, _ = fixture_params(fx)
    sol = make_solver(S, fx, st)
    checked = overflowed = 0
    per_agent = fixture_agent_params(fx)
    n_agents = len(st['radius'])
    maxn_of = per_agent.get('max_neighbors', np.full(n_agents, params.get('max_neighbors', 16)))
    nd_of = per_agent.get('neighbor_dist', np.full(n_agents, params.get('neighbor_dist', 10.0)))
    for t in range(len(fx['step'])):
        sol.set_state(fx['pos'][t], fx['vel'][t], fx['heading'][t], fx['flags'][t], fx['total_dist'][t])
        sol.set_vpref(fx['vpref'][t], st['vpref_mode'])
        try:
            sol.policy_pass(S.NBR_GRID)
        except S.ScaError as e:
            assert 'SCA_NBR_GRID needs' in str(e) and nd_of.max() < 4.0, str(e)
            sol.close()
            return
        nb, dg = sol.neighbors(), sol.diag()
        valid = fx['nbr_valid'][t].astype(bool)
        rows = np.nonzero(valid & ((dg['status'] & 32) == 0))[0]
        for i in rows:
            k = int(fx['nbr_n'][t][i])
            assert nb['nbr_n'][i] == k, (name, t, i)
            got = sorted(zip(nb['nbr_dsq'][i][:k].round(9), nb['nbr_kind'][i][:k], nb['nbr_id'][i][:k]))
            ref = sorted(zip(fx['nbr_dsq'][t][i][:k].round(9), fx['nbr_kind'][t][i][:k], fx['nbr_id'][t][i][:k]))
            assert got == ref, (name, t, i)
            checked += 1
        # overflowing lists: the max_neighbors nearest, ascending, all inside neighbor_dist (the reference's own list is visit-order dependent)
        for i in np.nonzero(valid & ((dg['status'] & 32) != 0) & (fx['coll_after_policy'][t] == 0))[0]:
            maxn = int(maxn_of[i])
            d = nb['nbr_dsq'][i][:maxn]
            assert nb['nbr_n'][i] == maxn and (np.diff(d) >= 0).all() and d[-1] < nd_of[i] ** 2, (name, t, i)
            assert d[0] <= fx['nbr_dsq'][t][i][0], (name, t, i)   # nothing nearer was missed
            overflowed += 1
    assert checked + overflowed > 0
    sol.close()
