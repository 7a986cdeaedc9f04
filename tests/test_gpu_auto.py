"""SCA_NBR_AUTO (round 4): the grid query for every agent + the kd query for the agents whose list the grid cannot give exactly (more
than max_neighbors objects in range, or two objects of one kind at the same rounded distance), the kd-tree still built every step --
beside the grid build and query, and in resident runs already behind the previous step's integrate stage.  Everything must equal
SCA_NBR_KDTREE: states, action rows, the neighbour lists entry for entry, the carried permutation.  (The recorded reference episodes
run through it in tests/test_gpu_parity.py::test_policy_pass_vs_golden[auto].)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(kind, n, seed):
    from sca_amd import scenarios, solver as S
    rng = np.random.default_rng(seed)
    if kind == 'circle':
        sc = scenarios.circle(n)
        pol = np.zeros(n, np.uint8)
    elif kind == 'cube':                                         # every policy incl. the Official LP (plane order = list order)
        sc = scenarios.random_cube(n, seed=seed)
        pol = (np.arange(n) % 5).astype(np.uint8)
    elif kind == 'lattice':                                      # integer coordinates: equal distances everywhere -> everybody is handed over
        sc = scenarios.random_cube(n, seed=seed)
        side = int(np.ceil(n ** (1 / 3)))
        idx = np.arange(n)
        sc['start'][:, 0] = 3.0 * (idx % side); sc['start'][:, 1] = 3.0 * ((idx // side) % side); sc['start'][:, 2] = 40.0 + 3.0 * (idx // (side * side))
        sc['goal'][:, :3] = sc['start'][rng.permutation(n), :3]
        pol = np.where(idx % 3 == 0, 4, idx % 4).astype(np.uint8)
    elif kind == 'dense':                                        # ~40 agents within neighborDist: overflow, collisions, fallback agents
        sc = scenarios.random_cube(n, seed=seed)
        sc['start'][:, :3] = rng.uniform(-9, 9, (n, 3)) + np.array([0, 0, 30.0])
        sc['goal'][:, :3] = rng.uniform(-9, 9, (n, 3)) + np.array([0, 0, 30.0])
        pol = (np.arange(n) % 5).astype(np.uint8)
    else:
        sc = scenarios.takeoff_landing(n)
        n = len(sc['start'])
        pol = np.where(np.arange(n) % 2 == 0, 0, 2).astype(np.uint8)
    return sc, pol, len(sc['start'])


def _solver(sc, pol, n, tracked):
    from sca_amd import scenarios, solver as S
    sol = S.BatchedSolver(max_agents=n, max_obstacles=max(1, len(sc['obs_radius'])))
    sol.set_obstacles(sc['obs_pos'], sc['obs_radius'])
    sol.set_agents(np.full(n, 0.5), np.ones(n), sc['goal'][:, :3], pol, S.zaxis_flags(sc['start'], sc['goal']),
                   scenarios.max_run_dist(sc['start'], sc['goal']))
    sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
    if tracked:
        sol.device_tracker_enable(sc['goal'][:, 3:6])
    return sol


@pytest.mark.parametrize('kind,n,tracked,steps,burst', [
    ('cube', 4096, False, 40, 1), ('cube', 4096, False, 40, 8), ('circle', 3000, True, 30, 5), ('circle', 20000, True, 16, 4),
    ('lattice', 3375, False, 24, 6), ('dense', 600, False, 20, 4), ('takeoff', 1024, True, 30, 6), ('circle', 100000, False, 8, 4),
    ('circle', 100000, True, 8, 4)])
def test_auto_mode_equals_kdtree_mode(kind, n, tracked, steps, burst):
    """side by side from the start state, `burst` steps per library call (burst > 1: the next pass's tree is built ahead, behind the
    integrate stage; burst = 1: every pass builds its own)"""
    from sca_amd import solver as S
    sc, pol, n = _scene(kind, n, seed=7)
    a, b = _solver(sc, pol, n, tracked), _solver(sc, pol, n, tracked)
    for t in range(0, steps, burst):
        a.run_steps(burst, S.NBR_KDTREE); b.run_steps(burst, S.NBR_AUTO)
        a.synchronize(); b.synchronize()
        sa, sb = a.get_state(), b.get_state()
        for k in ('pos', 'vel', 'heading', 'flags', 'total_dist', 'step_num'):
            assert np.array_equal(sa[k], sb[k]), (kind, n, t, k, int((sa[k] != sb[k]).sum()))
        assert np.array_equal(a.actions(), b.actions()), (kind, t)
        na, nb = a.neighbors(), b.neighbors()
        for k in ('nbr_valid', 'nbr_n', 'nbr_id', 'nbr_kind', 'nbr_dsq'):
            assert np.array_equal(na[k], nb[k]), (kind, n, t, k)
        assert np.array_equal(a.get_kd_perm(), b.get_kd_perm()), (kind, t)
        da, db = a.diag(), b.diag()
        assert np.array_equal(da['diag'], db['diag']) and np.array_equal(da['status'], db['status']), (kind, t)
    a.close(); b.close()


def test_auto_mode_hands_ties_and_overflows_to_the_kd_query():
    """the lattice scene has equal distances in every list and the dense one more than 16 objects in range: lists in the kd-tree's
    order all the same (the plain grid mode orders ties by id and flags overflows)"""
    from sca_amd import solver as S
    for kind, n in (('lattice', 1000), ('dense', 400)):
        sc, pol, n = _scene(kind, n, seed=3)
        a, b, g = _solver(sc, pol, n, False), _solver(sc, pol, n, False), _solver(sc, pol, n, False)
        for sol, mode in ((a, S.NBR_KDTREE), (b, S.NBR_AUTO), (g, S.NBR_GRID)):
            sol.run_steps(3, mode); sol.synchronize()
        na, nb, ng = a.neighbors(), b.neighbors(), g.neighbors()
        assert np.array_equal(na['nbr_id'], nb['nbr_id']) and np.array_equal(na['nbr_dsq'], nb['nbr_dsq'])
        assert not np.array_equal(na['nbr_id'], ng['nbr_id']), kind      # (the scene does what it is for: the grid alone differs)
        for s in (a, b, g):
            s.close()


@pytest.mark.parametrize('kind,n,look_every', [('cube', 4096, 1), ('cube', 4096, 5), ('lattice', 1000, 3), ('dense', 600, 1), ('dense', 600, 4)])
def test_env_step_leaves_the_kd_stream_unjoined_and_every_reader_joins(kind, n, look_every):
    """sca_env_step (the drop-in loop's one call per step) returns when the context's stream is through: an SCA_NBR_AUTO pass may still
    have its tree build and kd query on kd_stream.  The next sca_env_step copes by itself; any other entry point joins first -- so
    whatever is read between steps (every step, or only now and then with unjoined steps in between) equals the kd-tree mode's."""
    from sca_amd import solver as S
    sc, pol, n = _scene(kind, n, seed=11)
    a, b = _solver(sc, pol, n, False), _solver(sc, pol, n, False)
    for t in range(24):
        ra, rb = a.env_step(S.NBR_KDTREE), b.env_step(S.NBR_AUTO)
        assert ra == rb, (kind, t, ra, rb)                                 # the count of agents still under way
        if (t + 1) % look_every:
            continue
        # a different reader first each time: each of them must join by itself
        readers = [lambda s: s.get_kd_perm(), lambda s: s.neighbors()['nbr_id'], lambda s: s.get_state()['pos'], lambda s: s.actions()]
        first = readers[(t // look_every) % len(readers)]
        assert np.array_equal(first(a), first(b)), (kind, t)
        na, nb = a.neighbors(), b.neighbors()
        for k in ('nbr_valid', 'nbr_n', 'nbr_id', 'nbr_kind', 'nbr_dsq'):
            assert np.array_equal(na[k], nb[k]), (kind, t, k)
        sa, sb = a.get_state(), b.get_state()
        for k in ('pos', 'vel', 'heading', 'flags', 'total_dist', 'step_num'):
            assert np.array_equal(sa[k], sb[k]), (kind, t, k)
        assert np.array_equal(a.get_kd_perm(), b.get_kd_perm()), (kind, t)
    # a burst of resident steps behind unjoined single steps, and single steps behind a burst
    b.env_step(S.NBR_AUTO); a.env_step(S.NBR_KDTREE)
    a.run_steps(5, S.NBR_KDTREE); b.run_steps(5, S.NBR_AUTO)
    a.env_step(S.NBR_KDTREE); b.env_step(S.NBR_AUTO)
    assert np.array_equal(a.get_kd_perm(), b.get_kd_perm()) and np.array_equal(a.get_state()['pos'], b.get_state()['pos'])
    assert np.array_equal(a.neighbors()['nbr_id'], b.neighbors()['nbr_id'])
    a.close(); b.close()


@pytest.mark.parametrize('kind,n,tracked,mode', [('circle', 1024, True, 'kd'), ('takeoff', 1024, True, 'kd'), ('cube', 4096, False, 'auto'),
                                                  ('dense', 600, False, 'auto'), ('circle', 20000, True, 'kd')])
def test_events_as_stop_events_change_no_bit(kind, n, tracked, mode, monkeypatch):
    """The hand-over, fork and join events ride on the kernels they follow (hipExtLaunchKernelGGL stop events, the default) or are recorded
    behind them (SCA_EXT_STOP=0, read at sca_create): the same dependencies either way, so the same bits -- bursts of several steps (the
    fork rides on the previous step's last kernel, the moved positions' event on k_action) and single steps."""
    from sca_amd import solver as S
    sc, pol, n = _scene(kind, n, seed=5)
    m = S.NBR_AUTO if mode == 'auto' else S.NBR_KDTREE
    monkeypatch.setenv('SCA_EXT_STOP', '0')
    a = _solver(sc, pol, n, tracked)
    monkeypatch.delenv('SCA_EXT_STOP')
    b = _solver(sc, pol, n, tracked)
    for burst in (1, 6, 1, 5, 3):
        a.run_steps(burst, m); b.run_steps(burst, m)
        a.synchronize(); b.synchronize()
        sa, sb = a.get_state(), b.get_state()
        for k in ('pos', 'vel', 'heading', 'flags', 'total_dist', 'step_num'):
            assert np.array_equal(sa[k], sb[k]), (kind, burst, k)
        assert np.array_equal(a.actions(), b.actions()) and np.array_equal(a.get_kd_perm(), b.get_kd_perm()), (kind, burst)
        assert np.array_equal(a.neighbors()['nbr_id'], b.neighbors()['nbr_id']), (kind, burst)
    a.close(); b.close()


@pytest.mark.parametrize('kind,n,burst,div', [('dense', 600, 4, 1), ('lattice', 1000, 6, 1), ('dense', 600, 1, 1), ('cube', 4096, 8, 8), ('lattice', 3375, 1, 2)])
def test_tail_form_answers_the_listed_agents(kind, n, burst, div, monkeypatch):
    """Round 6: the kd query of the agents the grid query lists is answered by that query's own last workgroup, from the tree its
    pass's build publishes (KdTail, sca_kdbuild.hip.h), instead of by a launch of its own behind a stream wait -- once the list lengths
    that come back are known and small.  Forced here for ANY
    length (SCA_AUTO_TAIL_MAX) on scenes where hundreds are listed -- dense: more than 16 in range; lattice: ties everywhere, "too many
    for a list" = everybody -- with the back-off to the plain kd pass moved out of the way (SCA_AUTO_BACKOFF_DIV): everything must equal
    SCA_NBR_KDTREE, bursts (builds enqueued ahead) and single steps, and the passes must really have taken the tail."""
    from sca_amd import solver as S
    monkeypatch.setenv('SCA_AUTO_TAIL_MAX', str(1 << 30))
    monkeypatch.setenv('SCA_AUTO_BACKOFF_DIV', str(div))
    sc, pol, n = _scene(kind, n, seed=7)
    a, b = _solver(sc, pol, n, False), _solver(sc, pol, n, False)
    tails = 0
    for t in range(0, 36, burst):
        a.run_steps(burst, S.NBR_KDTREE); b.run_steps(burst, S.NBR_AUTO)
        a.synchronize(); b.synchronize()
        tails += bool(b.pass_forms() & S.FORM_AUTO_TAIL)
        sa, sb = a.get_state(), b.get_state()
        for k in ('pos', 'vel', 'heading', 'flags', 'total_dist', 'step_num'):
            assert np.array_equal(sa[k], sb[k]), (kind, n, t, k, int((sa[k] != sb[k]).sum()))
        assert np.array_equal(a.actions(), b.actions()), (kind, t)
        na, nb = a.neighbors(), b.neighbors()
        for k in ('nbr_valid', 'nbr_n', 'nbr_id', 'nbr_kind', 'nbr_dsq'):
            assert np.array_equal(na[k], nb[k]), (kind, n, t, k)
        assert np.array_equal(a.get_kd_perm(), b.get_kd_perm()), (kind, t)
        assert np.array_equal(a.diag()['status'], b.diag()['status']), (kind, t)
    st = b.auto_stats()
    assert tails >= 2, (kind, tails, st)                                   # (the first passes run the launch form: no count has come back yet)
    if kind != 'cube':
        assert st['passes_with_a_list_frac'] > 0 and st['listed_per_pass_max'] > 0, st       # somebody WAS listed while the tail form ran
    a.close(); b.close()


def test_tail_form_off_equals_on(monkeypatch):
    """SCA_AUTO_NO_TAIL=1 (the launch form on every pass) against the default on c3's own scene: the same bits, and the default takes the tail"""
    from sca_amd import solver as S
    sc, pol, n = _scene('cube', 4096, seed=2)
    monkeypatch.setenv('SCA_AUTO_NO_TAIL', '1')
    a = _solver(sc, np.full(n, 3, np.uint8), n, False)
    monkeypatch.delenv('SCA_AUTO_NO_TAIL')
    b = _solver(sc, np.full(n, 3, np.uint8), n, False)
    seen = 0
    for burst in (1, 7, 1, 10, 3, 12):
        a.run_steps(burst, S.NBR_AUTO); b.run_steps(burst, S.NBR_AUTO)
        a.synchronize(); b.synchronize()
        assert not (a.pass_forms() & S.FORM_AUTO_TAIL)
        seen += bool(b.pass_forms() & S.FORM_AUTO_TAIL)
        sa, sb = a.get_state(), b.get_state()
        for k in ('pos', 'vel', 'heading', 'flags', 'total_dist', 'step_num'):
            assert np.array_equal(sa[k], sb[k]), (burst, k)
        assert np.array_equal(a.get_kd_perm(), b.get_kd_perm()) and np.array_equal(a.neighbors()['nbr_id'], b.neighbors()['nbr_id']), burst
    assert seen >= 3, seen
    a.close(); b.close()


@pytest.mark.parametrize('tail_max,burst', [(330, 5), (310, 1), (345, 9)])
def test_form_switches_mid_episode(tail_max, burst, monkeypatch):
    """The host picks the launch-free form from list lengths that come back every fourth pass: in this dense blob 250 .. 445 agents
    are listed per pass, and a threshold in the middle of that range makes the forms alternate all episode long.  Mixed sequences of launch-form and launch-free
    passes -- their lists alternate by parity, the launch form's query runs on kd_stream, the other inside the pass's grid query --
    must stay equal to the kd-tree mode step for step."""
    from sca_amd import solver as S
    monkeypatch.setenv('SCA_AUTO_TAIL_MAX', str(tail_max))
    monkeypatch.setenv('SCA_AUTO_BACKOFF_DIV', '1')
    sc, pol, n = _scene('dense', 700, seed=23)
    a, b = _solver(sc, pol, n, False), _solver(sc, pol, n, False)
    seen = {True: 0, False: 0}
    for t in range(0, 180, burst):
        a.run_steps(burst, S.NBR_KDTREE); b.run_steps(burst, S.NBR_AUTO)
        a.synchronize(); b.synchronize()
        seen[bool(b.pass_forms() & S.FORM_AUTO_TAIL)] += 1
        sa, sb = a.get_state(), b.get_state()
        for k in ('pos', 'vel', 'heading', 'flags', 'total_dist', 'step_num'):
            assert np.array_equal(sa[k], sb[k]), (tail_max, t, k, int((sa[k] != sb[k]).sum()))
        na, nb = a.neighbors(), b.neighbors()
        for k in ('nbr_valid', 'nbr_n', 'nbr_id', 'nbr_kind', 'nbr_dsq'):
            assert np.array_equal(na[k], nb[k]), (tail_max, t, k)
        assert np.array_equal(a.get_kd_perm(), b.get_kd_perm()), (tail_max, t)
    print('passes seen in the launch-free / launch form:', seen, b.auto_stats())
    assert seen[True] > 0 and seen[False] > 0, seen           # both forms ran in this episode
    a.close(); b.close()
