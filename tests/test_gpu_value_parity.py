"""SCA AS SHIPPED at the BASELINE geometries -- c2 (N = 1024 circle), c5 (N = 16 384 take-off / landing, SCA + S-RVO3D) and c4
(N = 100 000 circle: 40-km Dubins paths, ~97 % of the agents re-planning every step) -- against the bit-exact path.

Two solvers step the same scene side by side:
  A  the product as bench.py's `value` leg runs it: the tracker inside every resident step (k_track_replan / k_replan* /
     k_track beside the kd build, the split solve, ...), state never leaves the device between steps;
  B  the native HOST tracker (sca_tracker_vpref: glibc's libm, pinned bit for bit to the reference's recorded v_pref,
     tests/test_tracker.py) feeding sca_set_vpref, one step at a time.
After every step: positions, velocities, headings, flags, the v_pref the pass used and the re-plan counters must be EQUAL.
The velocities are the metric's `v_new`; B's are the reference's given its v_pref rule (every solver test), so A == B is the
value leg's max |v_new - v_ref| = 0 on these scenes.  (Round 2 could only say "within 2e-5 on 99.9 % of the steps" and only
on N <= 100 episodes: the device ran on another libm.  See sca_glibc_math.h.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(kind, n):
    from sca_amd import scenarios, solver as S
    if kind == 'circle':
        sc = scenarios.circle(n)
        policy = np.zeros(n, np.uint8)
    else:
        sc = scenarios.takeoff_landing(n)
        n = len(sc['start'])
        policy = np.where(np.arange(n) % 2 == 0, 0, 2).astype(np.uint8)
    return dict(n=n, sc=sc, policy=policy, zaxis=S.zaxis_flags(sc['start'], sc['goal']), mrd=scenarios.max_run_dist(sc['start'], sc['goal']))


def _solver(scene):
    from sca_amd import solver as S
    sc, n = scene['sc'], scene['n']
    sol = S.BatchedSolver(max_agents=n, max_obstacles=max(1, len(sc['obs_radius'])))
    sol.set_obstacles(sc['obs_pos'], sc['obs_radius'])
    sol.set_agents(np.full(n, 0.5), np.ones(n), sc['goal'][:, :3], scene['policy'], scene['zaxis'], scene['mrd'])
    sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
    return sol


# (the 160-agent circle runs to the end of the episode: path lengths from 21 turning radii down to arrival -- the lean search's
# far block, its table pieces, the literal way for near problems, and the hand-over between them)
# ... and so do BASELINE config 5 itself (round 4: 40 steps of ~285 before): take-off / landing N = 16 384 until the last agent is
# done -- arrivals, the NEAR_GOAL hand-over, agents standing at their goals as obstacles-to-be, with the tracker on the device -- and
# BASELINE config 2 (N = 1024 circle) for 12 000 steps: the crossing in the middle, arrivals, collisions (~97 % of the agents are done by
# then; the last two or three dozen circle each other until the 3 x distance time-out of agent.py:74 at step ~12 400 -- how many exactly
# moved with round 6's bit-exact integration: 22 before, 31 with the device's sin / cos replaced by the restated glibc's)
@pytest.mark.parametrize('kind,n,steps', [('circle', 1024, 60), ('takeoff', 16384, 40), ('circle', 100000, 12), ('circle', 160, 800),
                                          ('takeoff', 16384, -400), ('circle', 1024, -12000)])
def test_value_leg_equals_host_tracker_run(kind, n, steps):
    to_the_end = steps < 0
    steps = abs(steps)
    from sca_amd import tracker
    scene = _scene(kind, n)
    sc, n = scene['sc'], scene['n']
    ext = np.isin(scene['policy'], (0, 5))
    a = _solver(scene)
    a.device_tracker_enable(sc['goal'][:, 3:6], in_pass=True)
    b = _solver(scene)
    host = tracker.DubinsTracker(sc['goal'][:, :3], sc['goal'][:, 3:6], np.ones(n), scene['zaxis'])
    agent_steps = replans_seen = 0
    for t in range(steps):
        st = b.get_state()
        active = ((st['flags'] & 7) == 0) & ext
        hv = np.nan_to_num(host.vpref(st['pos'], st['vel'], st['heading'], active.astype(np.uint8)))
        b.set_vpref(hv, ext.astype(np.uint8))
        b.run_steps(1)
        host.note_nbr0(b.nbr0())
        a.run_steps(1)
        a.synchronize(); b.synchronize()
        sa, sb = a.get_state(), b.get_state()
        for k in ('pos', 'vel', 'heading', 'flags', 'total_dist', 'step_num'):
            assert np.array_equal(sa[k], sb[k]), (kind, n, t, k, int((sa[k] != sb[k]).sum()))
        va, vb = a.diag()['vpref'], b.diag()['vpref']
        assert np.array_equal(va[active], vb[active]), (kind, n, t, float(np.abs(va[active] - vb[active]).max()))
        agent_steps += int(((st['flags'] & 7) == 0).sum())
        if to_the_end and not ((sb['flags'] & 7) == 0).any():
            steps = t + 1
            break
    if to_the_end:
        fl = b.get_state()['flags']
        assert ((fl & 7) != 0).mean() > 0.95, f'episode not (nearly) over after {steps} steps: {int(((fl & 7) == 0).sum())} agents still flying'
        assert (fl & 1).mean() > 0.9, 'fewer than 90 % of the agents arrived'
    ra, rb = a.device_tracker_replans()[ext], host.replans()[ext]
    assert np.array_equal(ra, rb)
    replans_seen = int(ra.sum())
    print(f'{kind} N={n}: {steps} steps, {agent_steps} agent-steps, {replans_seen} re-plans: identical')
    assert replans_seen >= int(ext.sum())                # everybody planned at least once
    host.close(); a.close(); b.close()
