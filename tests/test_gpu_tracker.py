"""The v_pref tracker on the device (sca_amd/csrc/sca_tracker.hip.h) against the golden v_pref of whole
reference episodes and against the native host tracker.

The device tracker is sca_dubins.hpp compiled for gfx950: the same statements as the host tracker and -- since round 3 -- the
same libm (sca_glibc_math.h: glibc's sin / cos / atan2 / acos / pow restated operation for operation; tests/test_gpu_glibc_math.py
checks the device build against the host's bit for bit).  So everything here is EQUALITY: v_pref of every agent-step, every
follow-or-re-plan decision, the velocities picked from them.  (Rounds 1-2 ran the device planner on the ROCm device library,
one ulp away from glibc in a few per cent of the calls: 0.1 % of the agent-steps then got a v_pref one or two 5-decimal steps
away, because the planner's radius search ends on comparisons of nearly equal path lengths.)
"""
import numpy as np
import pytest

from golden_util import fixture_agent_params, fixture_params, fixture_tracker_agent_params, load, static_inputs, tracked_param_fixtures

pytestmark = pytest.mark.gpu

EPISODES = ['F1_sca_circle8', 'F2_sca_circle100', 'F2_rvodubins_circle100', 'F4_sca_takeoff16', 'F4_mixed_takeoff16', 'F10_sca_exp3_map',
            'F13_fuzz_track_00', 'F13_fuzz_track_01', 'F13_fuzz_track_02', 'F13_fuzz_track_03',
            'F15_sca_circle1024']             # BASELINE config 2 itself, steps 0-3 stepped by the reference (2038 plans of 412 m)
EPISODES += tracked_param_fixtures()  # F16: turning_radius 0.8 / 2 / 3 / 10, pitchlims +-pi/6, (-0.5, 0.9), ..., neighborDist 1.5 ... 30


def _solver_for(fx, st, in_pass):
    from sca_amd import solver as S
    n = len(st['radius'])
    params, trk = fixture_params(fx)
    sol = S.BatchedSolver(max_agents=n, max_obstacles=max(1, len(st['obs_radius'])), params=params)
    sol.set_obstacles(st['obs_pos'], st['obs_radius'])
    sol.set_agents(st['radius'], st['pref_speed'], fx['goal'][0], st['policy'], st['zaxis'], st['max_run_dist'])
    if fixture_agent_params(fx):                                   # F17: attributes that differ from agent to agent
        sol.set_agent_params(**fixture_agent_params(fx))
    sol.device_tracker_enable(fx['goal6'][:, 3:6], in_pass=in_pass, **trk)
    if fixture_tracker_agent_params(fx):                           # F18: every agent its own turning radius and pitch limits (classes on the device)
        sol.device_tracker_set_agent_params(**fixture_tracker_agent_params(fx))
    return sol


@pytest.mark.parametrize('name', EPISODES)
def test_device_tracker_reproduces_reference_v_pref(name):
    """Open loop on the solver (states and agent.neighbors[0] come from the fixture), closed loop on the tracker's own
    records: every compute_v_pref of the episode, all re-plans included, bit for bit the reference's V_des."""
    fx = load(name)
    st = static_inputs(fx)
    n = len(st['radius'])
    ext = st['vpref_mode'].astype(bool)
    sol = _solver_for(fx, st, in_pass=False)
    nb0 = np.full(n, -1.0)
    total = 0
    for t in range(len(fx['step'])):
        called = fx['called'][t].astype(bool)
        active = called & ext
        sol.set_state(fx['pos'][t], fx['vel'][t], fx['heading'][t], np.where(called, 0, 1).astype(np.uint8))
        got = sol.device_tracker_vpref(nb0)
        want = fx['vpref'][t]
        assert np.array_equal(got[active], want[active]), (name, t, float(np.abs(got[active] - want[active]).max()))
        total += int(active.sum())
        v = fx['nbr_valid'][t].astype(bool)
        nb0[v] = np.where(fx['nbr_n'][t] > 0, fx['nbr_dsq'][t][:, 0], -1.0)[v]
    assert total > 0
    sol.close()


@pytest.mark.parametrize('name', EPISODES)
def test_sca_as_shipped_velocity_parity_with_device_tracker(name):
    """SCA as the reference ships it (v_pref from the Dubins tracker, scaPolicy.py:32,264-338) with the tracker ON THE
    DEVICE: new_velocity of every agent-step of the recorded episodes equals the reference's action bit for bit (15 589 tracked
    agent-steps over the ten episodes; round 2: 16 of them off by up to 2e-5).  Open loop on the states (they come from the
    fixture), closed loop on the tracker's records.  SCA_ST_TRACKER_EDGE is never set any more."""
    fx = load(name)
    st = static_inputs(fx)
    n = len(st['radius'])
    ext = st['vpref_mode'].astype(bool)
    sol = _solver_for(fx, st, in_pass=False)
    nb0 = np.full(n, -1.0)
    steps = 0
    for t in range(len(fx['step'])):
        called = fx['called'][t].astype(bool)
        sol.set_state(fx['pos'][t], fx['vel'][t], fx['heading'][t], fx['flags'][t], fx['total_dist'][t])
        sol.set_kd_perm(fx['perm'][t])
        sol.device_tracker_vpref(nb0)                       # compute_v_pref of every tracked agent, on the device
        sol.policy_pass()                                   # the pass reads that v_pref (tracker not re-run: in_pass=False)
        a = sol.actions()
        assert np.array_equal(a[called, :3], fx['action'][t][called, :3]), (name, t)
        assert not (sol.diag()['status'] & 64).any()
        steps += int((called & ext).sum())
        v = fx['nbr_valid'][t].astype(bool)
        nb0[v] = np.where(fx['nbr_n'][t] > 0, fx['nbr_dsq'][t][:, 0], -1.0)[v]
    assert steps > 0
    sol.close()


def _swarm(n, seed=0):
    from sca_amd import scenarios, solver as S
    sc = scenarios.random_cube(n, seed=seed)
    pol = np.where(np.arange(n) % 3 == 0, 3, 0).astype(np.uint8)          # SCA with every third agent ORCA3D (not tracked)
    sol = S.BatchedSolver(max_agents=n, max_obstacles=4)
    sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
    sol.set_agents(np.full(n, 0.5), np.ones(n), sc['goal'][:, :3], pol, S.zaxis_flags(sc['start'], sc['goal']),
                   scenarios.max_run_dist(sc['start'], sc['goal']))
    sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
    return sol, sc, pol


def test_in_pass_tracker_reads_the_previous_pass_neighbour_lists():
    """Resident stepping with the tracker inside every pass == stepping with the tracker called explicitly and fed
    agent.neighbors[0] through the host (sca_get_nbr0, the path the host tracker uses), bit for bit."""
    n, steps = 700, 25
    a, sc, _ = _swarm(n)
    a.device_tracker_enable(sc['goal'][:, 3:6], in_pass=True)
    a.run_steps(steps)
    a.synchronize()
    b, _, _ = _swarm(n)
    b.device_tracker_enable(sc['goal'][:, 3:6], in_pass=False)
    nb0 = np.full(n, -1.0)
    for _ in range(steps):
        b.device_tracker_vpref(nb0)
        b.run_steps(1)
        got = b.nbr0()
        nb0 = np.where(got > -2.0, got, nb0)
    sa, sb = a.get_state(), b.get_state()
    for k in ('pos', 'vel', 'heading', 'flags', 'total_dist'):
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(a.device_tracker_replans(), b.device_tracker_replans())
    assert a.device_tracker_replans().sum() >= n - n // 3                 # every tracked agent planned at least once
    a.close()
    b.close()


def test_device_tracker_feeds_the_solver_what_the_oracle_expects():
    """A pass with the device tracker: the action of every agent equals the oracle's on the v_pref the pass used, and the
    untracked (ORCA3D) agents still use the straight-line rule."""
    from oracle import oracle as orc
    from sca_amd import solver as S
    n = 1500
    sol, sc, pol = _swarm(n, seed=3)
    sol.device_tracker_enable(sc['goal'][:, 3:6])
    sol.run_steps(6)
    sol.synchronize()
    st = sol.get_state()
    perm = sol.get_kd_perm()
    sol.policy_pass()
    act = sol.actions()
    used = sol.diag()['vpref']
    ext = np.isin(pol, (0, 5))
    ref = orc.policy_step(st['pos'], st['vel'], st['heading'], np.full(n, 0.5), np.ones(n), st['flags'], sc['goal'][:, :3], pol,
                          S.zaxis_flags(sc['start'], sc['goal']), np.nan_to_num(used), ext.astype(np.uint8), perm,
                          np.zeros((0, 3)), np.zeros(0), nthreads=8)
    assert float(np.abs(act[:, :3] - ref['action'][:, :3]).max()) == 0.0
    sol.close()


def test_env_episode_with_device_tracker_equals_host_tracker_episode():
    """run_sca.py's default scene (16 drones on a circle, 8 obstacle spheres), whole episode through the drop-in API: the
    device tracker's episode IS the host tracker's -- same length, same final flags, the same positions at every step."""
    from sca_amd import env as E, scenarios, solver as S, tracker
    import math

    def run(device):
        sc = scenarios.circle(16, rad=10.0)
        agents = [E.Agent(start_pos=list(sc['start'][i]), goal_pos=list(sc['goal'][i]), vel=[0.0, 0.0, 0.0], radius=0.5,
                          pref_speed=1.0, policy=E.SCAPolicy, id=i) for i in range(16)]
        obstacles = [E.Obstacle(pos=[round(4.0 * math.cos(2 * j * math.pi / 8), 2), round(4.0 * math.sin(2 * j * math.pi / 8), 2), 5.0],
                                shape_dict={'shape': 'sphere', 'feature': 1.0}, id=j) for j in range(8)]
        fn = None if device else tracker.DubinsTracker(sc['goal'][:, :3], sc['goal'][:, 3:6], np.ones(16),
                                                       S.zaxis_flags(sc['start'], sc['goal']))
        env = E.MACAEnv(v_pref_fn=fn, device_tracker=device)
        env.set_agents(agents, obstacles=obstacles)
        steps = 0
        trail = []
        while steps < 2000:
            done = env.step({})
            steps += 1
            trail.append(env.pos.copy())
            if done:
                break
        return steps, env.flags.copy(), np.array(trail)

    s_host, f_host, p_host = run(False)
    s_dev, f_dev, p_dev = run(True)
    assert (f_host & 1).all() and not (f_host & 6).any()
    assert s_dev == s_host and np.array_equal(f_dev, f_host)
    assert np.array_equal(p_dev, p_host)


def test_kd_tail_launch_with_desynchronised_workgroups():
    """Regression: the last level launch of the kd build (k_kd_level_tail) lets every workgroup run depth-first through its
    own subtree, so workgroups are at different levels at the same time; table slots indexed per level parity then collide
    (a workgroup two levels down overwrote the chunk records later-starting workgroups had not read yet).  It only showed
    once the tracker's re-plans ran beside the build and delayed some of its workgroups: a converging 20 000-agent swarm
    failed around step 530.  Now: no failure, and the run equals the one with everything on one stream and the host-built
    tree, permutation included."""
    from sca_amd import scenarios, solver as S
    n = 20000
    rng = np.random.default_rng(5)
    sc = scenarios.random_cube(n, seed=1)
    c = sc['start'][:, :3].mean(0)
    goal = c + rng.normal(0, 3.0, (n, 3))
    goal[:, 2] = np.maximum(goal[:, 2], 5.0)
    outs = []
    for mode in (S.NBR_KDTREE, S.NBR_KDTREE_HOSTBUILD):
        sol = S.BatchedSolver(max_agents=n, max_obstacles=4)
        sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
        sol.set_agents(np.full(n, 0.5), np.ones(n), goal, np.zeros(n, np.uint8), S.zaxis_flags(sc['start'], sc['goal']),
                       np.full(n, 1e9))
        sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
        sol.device_tracker_enable(sc['goal'][:, 3:6])
        if mode == S.NBR_KDTREE:
            sol.run_steps(500, mode)
            sol.synchronize()                                  # raises if a build reported failure
        else:
            for _ in range(500):                               # the host build reads positions back: step by step
                sol.run_steps(1, mode)
            sol.synchronize()
        st = sol.get_state()
        outs.append((st['pos'].copy(), st['flags'].copy(), sol.get_kd_perm().copy(), sol.device_tracker_replans().copy()))
        if mode == S.NBR_KDTREE:
            sol.run_steps(200, mode)
            sol.synchronize()
        sol.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_device_tracker_on_random_scenes_vs_host_tracker():
    """Fuzz: 30 random scenes (SCA / RVO3D+Dubins among other policies, take-off agents, pitched starts and goals), 25 steps
    each; both trackers see the same states (the run follows the host tracker's v_pref): every v_pref component equal, re-plan
    counts equal."""
    from sca_amd import solver as S, tracker
    tot = diff = 0
    worst = 0.0
    for seed in range(30):
        rng = np.random.default_rng(50_000 + seed)
        n = int(rng.choice([5, 40, 150, 400, 1200]))
        side = float(rng.choice([8.0, 25.0, 80.0]))
        pos = rng.uniform(-side, side, (n, 3))
        pos[:, 2] = np.abs(pos[:, 2]) + rng.choice([0.0, 5.0])
        goal = rng.uniform(-side, side, (n, 3))
        goal[:, 2] = np.abs(goal[:, 2]) + 1.0
        if rng.random() < 0.4:
            goal[: n // 2, :2] = pos[: n // 2, :2]
        head = np.zeros((n, 3))
        head[:, 0] = rng.uniform(0, 2 * np.pi, n)
        head[:, 1] = rng.uniform(-0.6, 0.6, n)
        gh = np.zeros((n, 3))
        gh[:, 0] = rng.uniform(0, 2 * np.pi, n)
        gh[:, 1] = rng.uniform(-0.3, 0.3, n)
        policy = rng.choice([0, 0, 0, 5, 2, 3], n).astype(np.uint8)
        pref = rng.choice([1.0, 0.8, 1.5], n)
        zaxis = S.zaxis_flags(np.concatenate([pos, head], 1), np.concatenate([goal, gh], 1))
        ext = np.isin(policy, (0, 5))
        sol = S.BatchedSolver(max_agents=n, max_obstacles=1)
        sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
        sol.set_agents(np.full(n, 0.5), pref, goal, policy, zaxis, 3 * np.linalg.norm(pos - goal, axis=1) + 1)
        sol.set_state(pos, np.zeros((n, 3), np.float32), head, np.zeros(n, np.uint8))
        sol.device_tracker_enable(gh, in_pass=False)
        host = tracker.DubinsTracker(goal, gh, pref, zaxis, nthreads=8)
        nb0 = np.full(n, -1.0)
        for _ in range(25):
            st = sol.get_state()
            active = ((st['flags'] & 7) == 0) & ext
            hv = np.nan_to_num(host.vpref(st['pos'], st['vel'], st['heading'], active.astype(np.uint8)))
            dv = sol.device_tracker_vpref(nb0)
            d = np.abs(hv[active] - dv[active])
            if d.size:
                tot += d.size
                diff += int((d > 0).sum())
                worst = max(worst, float(d.max()))
            sol.set_vpref(hv, ext.astype(np.uint8))
            sol.run_steps(1)
            got = sol.nbr0()
            nb0 = np.where(got > -2.0, got, nb0)
            host.note_nbr0(got)
        assert np.array_equal(sol.device_tracker_replans()[ext], host.replans()[ext]), seed
        host.close()
        sol.close()
    assert tot > 100000
    assert diff == 0 and worst == 0.0, (diff, tot, worst)


def test_fused_tracker_kernel_equals_track_plus_replan(monkeypatch):
    """k_track_replan (decision + re-plan in one launch; opt-in since round 3: SCA_TRACKER_FUSE) against k_track + the re-plan
    kernel over the list ordered by expected search length (the default): whole resident episodes equal bit for bit, across
    the pass where the library switches from one form to the other, and the re-plan counters agree."""
    from sca_amd import scenarios, solver as S
    n = 50000                                                             # ~42 000 re-plans per pass: the lane-per-plan range
    sc = scenarios.circle(n)
    pol = np.where(np.arange(n) % 9 == 4, 2, 0).astype(np.uint8)          # SCA, every ninth agent S-RVO3D (not tracked)
    sols = []
    for nofuse in (True, False):
        if nofuse:
            monkeypatch.delenv('SCA_TRACKER_FUSE', raising=False)
        else:
            monkeypatch.setenv('SCA_TRACKER_FUSE', '1')
        sol = S.BatchedSolver(max_agents=n)
        sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
        sol.set_agents(np.full(n, 0.5), np.ones(n), sc['goal'][:, :3], pol, S.zaxis_flags(sc['start'], sc['goal']),
                       scenarios.max_run_dist(sc['start'], sc['goal']))
        sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
        sol.device_tracker_enable(sc['goal'][:, 3:6])                     # reads SCA_TRACKER_FUSE
        sols.append(sol)
    plain, fused = sols
    seen_fused = False
    for t in range(20):
        plain.run_steps(1); fused.run_steps(1)
        plain.synchronize(); fused.synchronize()
        assert not plain.pass_forms() & S.FORM_TRACK_FUSED
        seen_fused |= bool(fused.pass_forms() & S.FORM_TRACK_FUSED)
        a, b = plain.get_state(), fused.get_state()
        for k in ('pos', 'vel', 'heading', 'flags', 'total_dist', 'step_num'):
            assert np.array_equal(a[k], b[k]), (t, k)
        assert np.array_equal(plain.diag()['vpref'], fused.diag()['vpref'], equal_nan=True), t
        assert np.array_equal(plain.diag()['status'], fused.diag()['status']), t
    assert seen_fused
    assert np.array_equal(plain.device_tracker_replans(), fused.device_tracker_replans())
    plain.close(); fused.close()


@pytest.mark.parametrize('n,form', [(600, 'spec4'), (1500, 'spec3'), (3000, 'spec2'), (24000, 'quad'), (45000, 'lane')])
def test_replan_kernel_ranges_give_the_same_episode(n, form, monkeypatch):
    """The re-plan kernel is picked per pass by the pass's re-plan count (k_replan_group<64 / 32 / 16 / 4>: <= 1024 / 2048 /
    4096 / 32 768 re-plans, k_replan above).  Every one of them is the same planner: an episode with the ranges as they are
    equals the episode with the ranges moved so that another kernel does the work (SCA_TRK_SPEC*_MAX / SCA_TRK_MID_MAX), bit
    for bit."""
    from sca_amd import scenarios, solver as S
    sc = scenarios.circle(n)
    pol = np.zeros(n, np.uint8)
    keys = ('SCA_TRK_SPEC4_MAX', 'SCA_TRK_SPEC3_MAX', 'SCA_TRK_SPEC2_MAX', 'SCA_TRK_MID_MAX')
    other = {'spec4': ('0', '0', '0', '1000000'),          # -> the quad form
             'spec3': ('0', '0', '1000000', '1000000'),    # -> 16 lanes per plan
             'spec2': ('1000000', '1000000', '1000000', '1000000'),   # -> a wavefront per plan
             'quad': ('0', '0', '0', '1'),                 # -> one lane per plan
             'lane': ('0', '0', '0', '1000000')}[form]     # -> the quad form
    sols = []
    for moved in (False, True):
        for k, v in zip(keys, other):
            if moved:
                monkeypatch.setenv(k, v)
            else:
                monkeypatch.delenv(k, raising=False)
        sol = S.BatchedSolver(max_agents=n)
        sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
        sol.set_agents(np.full(n, 0.5), np.ones(n), sc['goal'][:, :3], pol, S.zaxis_flags(sc['start'], sc['goal']),
                       scenarios.max_run_dist(sc['start'], sc['goal']))
        sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
        sol.device_tracker_enable(sc['goal'][:, 3:6])
        sols.append(sol)
    a, b = sols
    for t in range(10):
        a.run_steps(1); b.run_steps(1)
        a.synchronize(); b.synchronize()
        sa, sb = a.get_state(), b.get_state()
        for k in ('pos', 'vel', 'heading', 'flags'):
            assert np.array_equal(sa[k], sb[k]), (form, t, k)
        assert np.array_equal(a.diag()['vpref'], b.diag()['vpref'], equal_nan=True), (form, t)
    assert np.array_equal(a.device_tracker_replans(), b.device_tracker_replans())
    a.close(); b.close()


@pytest.mark.parametrize('form', ['spec4', 'spec3', 'spec2', 'quad', 'lane'])
@pytest.mark.parametrize('family', ['c4', 'c2', 'c5', 'all'])
def test_device_planner_long_range_kats(family, form, monkeypatch):
    """tests/golden/F7b_dubins_kat_long.npz -- the REFERENCE's planner on c4 (4.7 .. 39.8 km), c2 (61 .. 412 m) and c5 (11 .. 17 m)
    poses -- through every re-plan kernel: k_replan_group<64 / 32 / 16 / 4> and the lane-per-plan k_replan (the c4 family alone fills
    its wavefront with far plans: the lean search's straight-line block; `all` mixes far and near lanes: the literal way).  Length,
    word, both radii, t / p of both maneuvers, sampling size and sample count of the record each kernel leaves: the reference's bits."""
    import ctypes as C
    from sca_amd import _lib, solver as S
    from test_dubins_kat_long import load_kats, words_of
    k = load_kats()
    sel = np.flatnonzero(np.char.startswith(k['family'], family.encode())) if family != 'all' else np.arange(len(k['length']))
    n = len(sel)
    env = {'spec4': None, 'spec3': ('0', None, None, None), 'spec2': ('0', '0', None, None), 'quad': ('0', '0', '0', None),
           'lane': ('0', '0', '0', '1')}[form]
    for key, v in zip(('SCA_TRK_SPEC4_MAX', 'SCA_TRK_SPEC3_MAX', 'SCA_TRK_SPEC2_MAX', 'SCA_TRK_MID_MAX'), env or (None,) * 4):
        if v is None:
            monkeypatch.delenv(key, raising=False)
        else:
            monkeypatch.setenv(key, v)
    qi, qf = k['qi'][sel], k['qf'][sel]
    sol = S.BatchedSolver(max_agents=n)
    sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
    start6 = np.concatenate([qi, np.zeros((n, 1))], 1)
    goal6 = np.concatenate([qf, np.zeros((n, 1))], 1)
    sol.set_agents(np.full(n, 0.5), np.ones(n), qf[:, :3], np.zeros(n, np.uint8), S.zaxis_flags(start6, goal6), np.full(n, 1e9))
    sol.set_state(qi[:, :3], np.zeros((n, 3), np.float32), start6[:, 3:6], np.zeros(n, np.uint8))
    sol.device_tracker_enable(goal6[:, 3:6], in_pass=False)
    sol.device_tracker_vpref(np.full(n, -1.0))                  # the first compute_v_pref of every agent: a plan each
    assert np.array_equal(sol.device_tracker_replans(), np.ones(n, np.int32))
    want_form = {'lane': S.FORM_REPLAN_LANE}.get(form, S.FORM_REPLAN_FEW)
    assert sol.pass_forms() & want_form, (form, sol.pass_forms())
    o = np.zeros(24)
    for j, i in enumerate(sel):
        assert sol.L.sca_device_tracker_debug(sol.ctx, j, _lib.ptr(o, C.c_double)) == 0
        assert o[8] == k['length'][i] and words_of(o) == k['mode'][i], (family, form, i, o[8], k['length'][i], words_of(o), k['mode'][i])
        assert o[0] == k['radii'][i, 0] and o[4] == k['radii'][i, 1], (family, form, i)
        assert o[1] == k['tpq'][i, 0] and o[2] == k['tpq'][i, 1] and o[5] == k['tpq'][i, 3] and o[6] == k['tpq'][i, 4], (family, form, i)
        assert o[9] == k['sampling'][i] and int(o[13]) == k['n'][i], (family, form, i)
    sol.close()


@pytest.mark.parametrize('form', ['spec4', 'spec3', 'spec2', 'quad', 'lane'])
@pytest.mark.parametrize('pset', range(6))
def test_device_planner_param_kats(pset, form, monkeypatch):
    """F7c (tests/test_dubins_kat_params.py): the reference's planner at Rmin 0.8 / 3 / 10 and pitch limits -+pi/4, -+pi/6, (-0.5, 0.9),
    (-0.2, 0.2) -- every re-plan kernel with sca_device_tracker_enable(..., turning_radius, pitch_lo, pitch_hi): length, word, radii,
    t / p, sampling size and sample count of the record each kernel leaves are the reference's bits (the speculation trees were fitted at
    Rmin = 1.5: a walk that leaves them must still end where the search ends)."""
    import ctypes as C
    from sca_amd import _lib, solver as S
    from test_dubins_kat_long import words_of
    from test_dubins_kat_params import load_kats
    k = load_kats()
    sel = np.flatnonzero(k['set'] == pset)
    n = len(sel)
    env = {'spec4': None, 'spec3': ('0', None, None, None), 'spec2': ('0', '0', None, None), 'quad': ('0', '0', '0', None),
           'lane': ('0', '0', '0', '1')}[form]
    for key, v in zip(('SCA_TRK_SPEC4_MAX', 'SCA_TRK_SPEC3_MAX', 'SCA_TRK_SPEC2_MAX', 'SCA_TRK_MID_MAX'), env or (None,) * 4):
        if v is None:
            monkeypatch.delenv(key, raising=False)
        else:
            monkeypatch.setenv(key, v)
    qi, qf = k['qi'][sel], k['qf'][sel]
    sol = S.BatchedSolver(max_agents=n)
    sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
    start6 = np.concatenate([qi, np.zeros((n, 1))], 1)
    goal6 = np.concatenate([qf, np.zeros((n, 1))], 1)
    sol.set_agents(np.full(n, 0.5), np.ones(n), qf[:, :3], np.zeros(n, np.uint8), S.zaxis_flags(start6, goal6), np.full(n, 1e9))
    sol.set_state(qi[:, :3], np.zeros((n, 3), np.float32), start6[:, 3:6], np.zeros(n, np.uint8))
    sol.device_tracker_enable(goal6[:, 3:6], turning_radius=float(k['set_rmin'][pset]), pitchlims=tuple(k['set_pitchlims'][pset]), in_pass=False)
    sol.device_tracker_vpref(np.full(n, -1.0))
    assert np.array_equal(sol.device_tracker_replans(), np.ones(n, np.int32))
    o = np.zeros(24)
    for j, i in enumerate(sel):
        assert sol.L.sca_device_tracker_debug(sol.ctx, j, _lib.ptr(o, C.c_double)) == 0
        assert o[8] == k['length'][i] and words_of(o) == k['mode'][i], (pset, form, i, o[8], k['length'][i], words_of(o), k['mode'][i])
        assert o[0] == k['radii'][i, 0] and o[4] == k['radii'][i, 1], (pset, form, i)
        assert o[1] == k['tpq'][i, 0] and o[2] == k['tpq'][i, 1] and o[5] == k['tpq'][i, 3] and o[6] == k['tpq'][i, 4], (pset, form, i)
        assert o[9] == k['sampling'][i] and int(o[13]) == k['n'][i], (pset, form, i)
    sol.close()


def test_group_fused_form_gives_the_same_episode(monkeypatch):
    """k_track_group (round 4: shards of <= 1024 agents take the follow-or-re-plan decision and the 64-lane search in ONE launch, a
    wavefront per agent) against k_track + k_replan_group<64> (SCA_TRACKER_NOGROUPFUSE=1): state, v_pref, re-plan counters equal after
    every step of a crossing with followers, re-planners, arrivals and agents of an untracked policy."""
    from sca_amd import scenarios, solver as S
    n = 700
    sc = scenarios.circle(n)
    pol = np.where(np.arange(n) % 7 == 3, 2, 0).astype(np.uint8)          # SCA, every seventh agent S-RVO3D (not tracked)
    sols = []
    for nofuse in ('1', None):
        if nofuse:
            monkeypatch.setenv('SCA_TRACKER_NOGROUPFUSE', nofuse)
        else:
            monkeypatch.delenv('SCA_TRACKER_NOGROUPFUSE', raising=False)
        sol = S.BatchedSolver(max_agents=n)
        sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
        sol.set_agents(np.full(n, 0.5), np.ones(n), sc['goal'][:, :3], pol, S.zaxis_flags(sc['start'], sc['goal']),
                       scenarios.max_run_dist(sc['start'], sc['goal']))
        sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
        sol.device_tracker_enable(sc['goal'][:, 3:6])                     # reads the switch
        sols.append(sol)
    plain, fused = sols
    for t in range(60):
        plain.run_steps(1); fused.run_steps(1)
        plain.synchronize(); fused.synchronize()
        assert not (plain.pass_forms() & S.FORM_TRACK_FUSED) and (fused.pass_forms() & S.FORM_TRACK_FUSED)
        a, b = plain.get_state(), fused.get_state()
        for k in ('pos', 'vel', 'heading', 'flags', 'total_dist', 'step_num'):
            assert np.array_equal(a[k], b[k]), (t, k)
        assert np.array_equal(plain.diag()['vpref'], fused.diag()['vpref'], equal_nan=True), t
    assert np.array_equal(plain.device_tracker_replans(), fused.device_tracker_replans())
    assert plain.device_tracker_replans().sum() > 10 * n
    plain.close(); fused.close()
