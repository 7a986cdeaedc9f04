"""Stress beyond the test suite (needs a GPU): random scenes of many sizes (64 .. 40 000 agents) and scales (8 m .. 20 km cubes; level
flight, height differences of 1e-14, random yaw / pitch offsets), the device tracker inside the resident step against the bit-exact host
tracker -- state, v_pref and re-plan counters must be equal after every step.   python tools/fuzz_track.py <seed> <scenes>"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from sca_amd import scenarios, solver as S, tracker
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
nscenes = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
plans = agent_steps = 0
t0 = time.time()
for sc_i in range(nscenes):
    n = int(rng.choice([64, 300, 1000, 3000, 9000, 20000, 40000]))
    scale = float(rng.choice([8.0, 30.0, 100.0, 400.0, 2000.0, 20000.0]))
    steps = int(rng.integers(8, 30))
    start = np.zeros((n, 6)); goal = np.zeros((n, 6))
    start[:, :3] = rng.uniform(-scale, scale, (n, 3)); goal[:, :3] = rng.uniform(-scale, scale, (n, 3))
    if rng.random() < 0.4: start[:, 2] = goal[:, 2] = 10.0                      # level
    if rng.random() < 0.3: goal[:, 2] = start[:, 2] + rng.choice([1e-14, -3e-13, 0.0, 2e-9], n)
    d = goal[:, :3] - start[:, :3]
    yaw = np.mod(np.arctan2(d[:, 1], d[:, 0]) + rng.normal(0, 0.4, n), 2 * np.pi)
    pitch = rng.choice([0.0, 1.0], n) * rng.normal(0, 0.2, n)
    start[:, 3] = yaw; start[:, 4] = pitch; goal[:, 3] = np.mod(yaw + rng.normal(0, 0.5, n), 2 * np.pi); goal[:, 4] = rng.choice([0.0, 1.0], n) * rng.normal(0, 0.1, n)
    policy = np.zeros(n, np.uint8)
    zaxis = S.zaxis_flags(start, goal); mrd = scenarios.max_run_dist(start, goal)
    def mk():
        sol = S.BatchedSolver(max_agents=n)
        sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
        sol.set_agents(np.full(n, 0.5), np.ones(n), goal[:, :3], policy, zaxis, mrd)
        sol.set_state(start[:, :3], np.zeros((n, 3), np.float32), start[:, 3:6], np.zeros(n, np.uint8))
        return sol
    a = mk(); a.device_tracker_enable(goal[:, 3:6], in_pass=True)
    b = mk()
    host = tracker.DubinsTracker(goal[:, :3], goal[:, 3:6], np.ones(n), zaxis)
    ok = True
    for t in range(steps):
        st = b.get_state()
        active = ((st['flags'] & 7) == 0)
        hv = np.nan_to_num(host.vpref(st['pos'], st['vel'], st['heading'], active.astype(np.uint8)))
        b.set_vpref(hv, np.ones(n, np.uint8))
        b.run_steps(1); host.note_nbr0(b.nbr0())
        a.run_steps(1); a.synchronize(); b.synchronize()
        sa, sb = a.get_state(), b.get_state()
        same = all(np.array_equal(sa[k], sb[k]) for k in ('pos', 'vel', 'heading', 'flags')) and np.array_equal(a.diag()['vpref'][active], b.diag()['vpref'][active])
        if not same:
            ok = False
            print('MISMATCH scene', sc_i, 'n', n, 'scale', scale, 'step', t, int((sa['pos'] != sb['pos']).any(axis=1).sum()), 'agents differ'); break
    if ok and not np.array_equal(a.device_tracker_replans(), host.replans()): ok = False; print('REPLAN COUNT MISMATCH scene', sc_i)
    bad += not ok
    plans += int(host.replans().sum()); agent_steps += n * steps
    host.close(); a.close(); b.close()
print('scenes', nscenes, 'bad', bad, 'agent-steps', agent_steps, 're-plans compared', plans, 'seconds %.0f' % (time.time() - t0))
