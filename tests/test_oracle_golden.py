"""CPU oracle (oracle/sca_oracle.c) pinned against golden vectors recorded from the reference itself
(tools/gen_golden.py).  Bit-exact unless stated."""
import numpy as np
import pytest

from golden_util import episode_fixtures, fixture_agent_params, fixture_params, load, static_inputs


def test_candidate_table(oracle):
    fx = load('F0_candidates')
    L = oracle.lib()
    for num_N in (256, 128):
        out = np.zeros((2 * num_N, 3))
        c = L.orc_candidate_table(1.0, num_N, oracle._d(out), 2 * num_N)
        assert c == 2 * num_N
        assert np.array_equal(out, fx[f'cand{num_N}'])


def test_kat_scalar_helpers(oracle):
    fx = load('F8_kat')
    L = oracle.lib()
    a, b = fx['a'], fx['b']
    n = len(a)
    v3 = oracle.vec3
    assert all(L.orc_l3norm(oracle._d(v3(a[i])), oracle._d(v3(b[i]))) == fx['l3'][i] for i in range(n))
    assert all(L.orc_l3normsq(oracle._d(v3(a[i])), oracle._d(v3(b[i]))) == fx['l3sq'][i] for i in range(n))
    assert all(L.orc_distance(oracle._d(v3(a[i])), oracle._d(v3(b[i]))) == fx['dist'][i] for i in range(n))
    v = fx['v']
    assert all(L.orc_get_phi(oracle._d(v3(v[i]))) == fx['phi'][i] for i in range(n))
    assert all(L.orc_pi_2_pi(float(x)) == y for x, y in zip(fx['ang'], fx['p2p']))
    assert all(L.orc_mod2pi(float(x)) == y for x, y in zip(fx['ang'], fx['m2p']))
    assert all(L.orc_trunc5(float(x)) == y for x, y in zip(fx['tr_in'], fx['tr']))
    vf = fx['vf']
    import ctypes as C
    for i in range(n):
        f = np.ascontiguousarray(vf[i], np.float32)
        assert L.orc_l3norm_mixed(oracle._d(v3(v[i])), oracle._p(f, C.c_float)) == fx['l3_mixed'][i]
        assert L.orc_l3norm_f32zero(oracle._p(f, C.c_float)) == fx['l3_f32zero'][i]
        assert L.orc_satisfied_constraint(oracle._p(f, C.c_float), float(fx['posz'][i]), oracle._d(v3(v[i]))) == fx['sat'][i]
    for i in range(n):
        r = L.orc_is_intersect(oracle._d(v3(fx['pA'][i])), oracle._d(v3(fx['pB'][i])), float(fx['R'][i]),
                               oracle._d(v3(fx['vd'][i])))
        assert r == fx['isx'][i]
    out = np.zeros(7)
    for i in range(n):
        L.orc_cartesian2spherical(oracle._d(v3(fx['head'][i])), oracle._d(v3(fx['vv'][i])), 0, oracle._d(out))
        assert np.array_equal(out, fx['c2s'][i])
        L.orc_cartesian2spherical(oracle._d(v3(fx['head'][i])), oracle._d(v3(fx['vv'][i])), 1, oracle._d(out))
        assert np.array_equal(out, fx['c2s_off'][i])


def test_round5_python_semantics(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(0, 500, 200000), np.arange(0, 2000) * 1e-5 + 5e-6, [0.285, 1.005, 2.675, 0.000005]])
    for x in xs:
        assert L.orc_round5_py(float(x)) == round(float(x), 5)


@pytest.fixture
def oracle_params(oracle):
    """oracle.set_params is process-wide: every test that sets the recorded parameters of an F16 fixture restores the defaults."""
    def apply(fx):
        oracle.set_params(**fixture_params(fx)[0])
        per_agent = fixture_agent_params(fx)                  # F17: attributes that differ from agent to agent
        oracle.set_agent_params(len(fx['radius']) if per_agent else 0, **per_agent)
    yield apply
    oracle.set_params()
    oracle.set_agent_params()


@pytest.mark.parametrize('name', episode_fixtures())
def test_policy_step_matches_reference(oracle, oracle_params, name):
    fx = load(name)
    oracle_params(fx)
    st = static_inputs(fx)
    T = len(fx['step'])
    for t in range(T):
        out = oracle.policy_step(fx['pos'][t], fx['vel'][t], fx['heading'][t], st['radius'], st['pref_speed'],
                                 fx['flags'][t], fx['goal'][t], st['policy'], st['zaxis'], fx['vpref'][t],
                                 st['vpref_mode'], fx['perm'][t], st['obs_pos'], st['obs_radius'], nthreads=4)
        called = fx['called'][t].astype(bool)
        assert np.array_equal(out['perm'], fx['perm_after'][t]), (name, t, 'kd permutation')
        # neighbour lists
        valid = fx['nbr_valid'][t].astype(bool)
        assert np.array_equal(out['nbr_valid'].astype(bool), valid), (name, t)
        assert np.array_equal(out['nbr_n'][valid], fx['nbr_n'][t][valid]), (name, t, 'nbr_n')
        assert np.array_equal(out['nbr_id'][valid], fx['nbr_id'][t][valid]), (name, t, 'nbr_id')
        assert np.array_equal(out['nbr_kind'][valid], fx['nbr_kind'][t][valid]), (name, t, 'nbr_kind')
        assert np.array_equal(out['nbr_dsq'][valid], fx['nbr_dsq'][t][valid]), (name, t, 'nbr_dsq')
        # straight-line v_pref (policies that compute it here)
        own = called & ~st['vpref_mode'].astype(bool)
        assert np.array_equal(out['vpref'][own], fx['vpref'][t][own]), (name, t, 'v_pref')
        # collision flag set inside insert*Neighbor
        assert np.array_equal((out['flags'] >> 1) & 1, fx['coll_after_policy'][t]), (name, t, 'collision')
        # decisions
        sel = fx['n_suit'][t] >= 0
        assert np.array_equal(out['diag'][sel, 0], fx['n_suit'][t][sel]), (name, t, 'n_suit')
        assert np.array_equal(out['diag'][sel, 1], fx['fallback'][t][sel]), (name, t, 'fallback')
        lp = fx['plane_fail'][t] >= 0
        assert np.array_equal(out['diag'][lp, 3], fx['plane_fail'][t][lp]), (name, t, 'planeFail')
        assert np.array_equal(out['diag'][lp, 4], fx['lp4'][t][lp]), (name, t, 'lp4')
        # the action row, bit-exact in float32 (mampenv.py:31,40)
        assert np.array_equal(out['action'], fx['action'][t]), (name, t, 'action',
                                                                np.abs(out['action'] - fx['action'][t]).max())
        assert not out['status'].any()


@pytest.mark.parametrize('name', episode_fixtures())
def test_env_update_matches_reference(oracle, oracle_params, name):
    fx = load(name)
    oracle_params(fx)
    st = static_inputs(fx)
    T = len(fx['step'])
    n = len(st['radius'])
    for t in range(T):
        flags_in = fx['flags'][t] | (fx['coll_after_policy'][t] << 1)
        out = oracle.env_update(fx['pos'][t], fx['vel'][t], fx['heading'][t], st['radius'], flags_in, fx['goal'][t],
                                fx['action'][t], fx['total_dist'][t], st['max_run_dist'], np.zeros(n, np.int32),
                                st['obs_pos'], st['obs_radius'])
        assert np.array_equal(out['pos'], fx['pos_after'][t]), (name, t, 'pos')
        assert np.array_equal(out['heading'], fx['heading_after'][t]), (name, t, 'heading')
        assert np.array_equal(out['vel'], fx['vel_after'][t]), (name, t, 'vel')
        assert np.array_equal(out['total_dist'], fx['total_dist_after'][t]), (name, t, 'total_dist')
        assert np.array_equal(out['flags'], fx['flags_after'][t]), (name, t, 'flags')
