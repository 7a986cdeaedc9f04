import sys, time, os, numpy as np
sys.path.insert(0, '.')
from sca_amd import scenarios, tracker as trk, solver as S
n = 100000
sc = scenarios.circle(n)
print('cpus', os.cpu_count())
pos = sc['start'][:, :3].copy(); head = sc['start'][:, 3:6].copy()
vel = np.zeros((n, 3), np.float32)
for nt in (16, 32, 64, 128, 256):
    tr = trk.DubinsTracker(sc['goal'][:, :3], sc['goal'][:, 3:6], np.ones(n), S.zaxis_flags(sc['start'], sc['goal']), nthreads=nt)
    act = np.ones(n, np.uint8)
    t0 = time.perf_counter(); v = tr.vpref(pos, vel, head, act); t1 = time.perf_counter()
    # second call: agents moved a little off the path -> typical re-plan pattern
    pos2 = pos + 0.1 * v + 0.003
    t2 = time.perf_counter(); v2 = tr.vpref(pos2, (0.9 * v).astype(np.float32), head, act); t3 = time.perf_counter()
    print('threads', nt, 'first call %.1f ms' % ((t1 - t0) * 1e3), 'second %.1f ms' % ((t3 - t2) * 1e3), 'replans', int(tr.replans().sum()))
