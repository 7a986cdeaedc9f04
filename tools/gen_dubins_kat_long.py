"""Golden-vector generator for the 3-D Dubins planner AT BASELINE GEOMETRY (runs only in the build container; imports
/root/reference).  Writes tests/golden/F7b_dubins_kat_long.npz:

  c4   N = 100 000 circle (paths of ~39.8 km, d = D / radius ~ 26 500 turning radii: the lean search's far block):
       start poses of agents spread over the ring, and perturbed mid-flight poses (somewhere along the chord to the goal,
       a few metres off it, yaw / pitch off the chord's direction)                           -- length, word, radii, t/p/q, 3 samples
  c2   N = 1024 circle (412 m): start poses and perturbed mid-flight poses                  -- the same + EVERY path sample
  c5   N = 16 384 take-off / landing cells (15-17 m, start and goal share x, y)             -- the same + EVERY path sample

The planner is the reference's own `dubinsmaneuver3d.dubinsmaneuver3d`, unpatched (a c4 plan takes it 7-13 s: `generate_course`
walks the 40-km path for every candidate radius although the 3-D planner never reads its output), called with the arguments
`compute_dubins` (scaPolicy.py:92-96) builds: qi = pos | heading, qf = goal | goal heading, Rmin = 1.5, pitchlims = -+pi/4.
Poses come from sca_amd/scenarios.py (the synthetic generalisations of run_sca.py's generators that bench.py uses)."""
import math
import os
import sys
from multiprocessing import Pool

sys.path.insert(0, '/root/reference')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import matplotlib

matplotlib.use('Agg')
import numpy as np

PL = [-math.pi / 4, math.pi / 4]
R = 1.5


def plan(case):
    fam, qi, qf, keep = case
    from mamp.policies.sca import dubinsmaneuver3d as d3
    m = d3.dubinsmaneuver3d(np.array(qi, dtype=np.float64), np.array(qf, dtype=np.float64), R, PL)
    h, v = m.maneuvers2d
    path = np.array(m.path, dtype=np.float64)
    n = len(path)
    return dict(fam=fam, qi=qi[:5], qf=qf[:5], length=float(m.length), mode=m.mode, n=n, radii=[float(h.r_min), float(v.r_min)],
                tpq=[float(h.t), float(h.p), float(h.q), float(v.t), float(v.p), float(v.q)], sampling=float(m.sampling_size),
                first=path[0], mid=path[n // 2], last=path[n - 1], samples=path if keep else None)


def midflight(rng, start, goal, frac, off, dyaw, dpitch):
    """a pose an agent could have `frac` of the way along its chord after avoiding somebody: `off` metres beside the chord,
    heading `dyaw` / `dpitch` off the direction it would fly"""
    p0, p1 = start[:3], goal[:3]
    chord = p1 - p0
    pos = p0 + frac * chord + rng.uniform(-off, off, 3)
    yaw = (math.atan2(chord[1], chord[0]) + rng.uniform(-dyaw, dyaw)) % (2 * math.pi)
    return np.concatenate([pos, [yaw, rng.uniform(-dpitch, dpitch), 0.0]])


def main():
    from sca_amd import scenarios
    rng = np.random.default_rng(74)
    cases = []
    c4 = scenarios.circle(100000)
    ids = rng.choice(100000, 18, replace=False)
    for i in ids:                                                # start poses: level, heading at the centre
        cases.append(('c4_start', c4['start'][i], c4['goal'][i], False))
    for i in rng.choice(100000, 30, replace=False):              # mid-flight, perturbed (incl. tiny height differences)
        q = midflight(rng, c4['start'][i], c4['goal'][i], rng.uniform(0.0005, 0.9), 3.0, 0.6, 0.3)
        if rng.random() < 0.3:
            q[2] = c4['goal'][i][2] + rng.uniform(-1, 1) * 10.0 ** rng.integers(-14, -2)     # dz ~ 1e-14 .. 1e-3 over 40 km
        cases.append(('c4_mid', q, c4['goal'][i], False))
    c2 = scenarios.circle(1024)
    for i in rng.choice(1024, 8, replace=False):
        cases.append(('c2_start', c2['start'][i], c2['goal'][i], True))
    for i in rng.choice(1024, 12, replace=False):
        cases.append(('c2_mid', midflight(rng, c2['start'][i], c2['goal'][i], rng.uniform(0.01, 0.9), 2.0, 0.8, 0.4), c2['goal'][i], True))
    c5 = scenarios.takeoff_landing(16384)
    for i in rng.choice(16384, 12, replace=False):
        cases.append(('c5_start', c5['start'][i], c5['goal'][i], True))
    for i in rng.choice(16384, 12, replace=False):               # on the way up / down, pushed aside by a neighbour
        s, g = c5['start'][i], c5['goal'][i]
        q = s.copy()
        q[:3] = s[:3] + rng.uniform(0.05, 0.9) * (g[:3] - s[:3]) + np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), 0.0])
        q[3] = rng.uniform(0, 2 * math.pi)
        q[4] = rng.uniform(-0.7, 0.7)
        cases.append(('c5_mid', q, g, True))
    cases = [(f, [float(x) for x in qi], [float(x) for x in qf], k) for f, qi, qf, k in cases]
    with Pool(int(os.environ.get('KAT_WORKERS', '6'))) as pool:
        res = pool.map(plan, cases, chunksize=1)
    K = len(res)
    off = np.zeros(K + 1, np.int64)
    chunks = []
    for k, r in enumerate(res):
        off[k + 1] = off[k] + (len(r['samples']) if r['samples'] is not None else 0)
        if r['samples'] is not None:
            chunks.append(r['samples'])
    out = dict(family=np.array([r['fam'] for r in res], dtype='S8'), qi=np.array([r['qi'] for r in res]), qf=np.array([r['qf'] for r in res]),
               rmin=np.float64(R), pitchlims=np.array(PL), length=np.array([r['length'] for r in res]),
               mode=np.array([r['mode'] for r in res], dtype='S6'), n=np.array([r['n'] for r in res], np.int32),
               radii=np.array([r['radii'] for r in res]), tpq=np.array([r['tpq'] for r in res]),
               sampling=np.array([r['sampling'] for r in res]), first=np.array([r['first'] for r in res]),
               mid=np.array([r['mid'] for r in res]), last=np.array([r['last'] for r in res]),
               samples_off=off, samples=np.concatenate(chunks) if chunks else np.zeros((0, 5)),
               numpy_version=np.array(np.__version__))
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'F7b_dubins_kat_long.npz'), **out)
    for fam in sorted(set(r['fam'] for r in res)):
        ls = [r['length'] for r in res if r['fam'] == fam]
        print(fam, len(ls), 'plans, length', min(ls), '..', max(ls), 'words', sorted(set(r['mode'] for r in res if r['fam'] == fam)))


if __name__ == '__main__':
    main()
