"""Golden-vector generator for the 3-D Dubins planner AWAY FROM THE DEFAULT Rmin / pitchlims (runs only in the build container; imports
/root/reference).  Writes tests/golden/F7c_dubins_kat_params.npz: plans by the reference's own `dubinsmaneuver3d.dubinsmaneuver3d`
(unpatched) called as `compute_dubins` calls it (scaPolicy.py:92-96) with `agent.turning_radius` in {0.8, 3.0, 10.0} and
`agent.pitchlims` in {-+pi/4, -+pi/6, (-0.5, 0.9), (-0.2, 0.2)} -- end points 1 .. 60 turning radii apart (both sides of the lean search's
d >= 7 block), level, climbing at and beyond the pitch limit (the planner must spiral), straight above each other (take-off), and
mid-flight poses with yaw / pitch off the chord.  Per plan: length (64 bits), word, radii, t / p / q of both 2-D maneuvers, sampling
size, sample count and EVERY path sample."""
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

SETS = [(0.8, (-math.pi / 4, math.pi / 4)), (3.0, (-math.pi / 6, math.pi / 6)), (10.0, (-0.5, 0.9)), (3.0, (-math.pi / 4, math.pi / 4)),
        (10.0, (-math.pi / 6, math.pi / 6)), (0.8, (-0.2, 0.2))]


def plan(case):
    s, qi, qf = case
    R, PL = SETS[s]
    from mamp.policies.sca import dubinsmaneuver3d as d3
    m = d3.dubinsmaneuver3d(np.array(qi, dtype=np.float64), np.array(qf, dtype=np.float64), R, list(PL))
    h, v = m.maneuvers2d
    path = np.array(m.path, dtype=np.float64)
    return dict(set=s, qi=qi[:5], qf=qf[:5], length=float(m.length), mode=m.mode, n=len(path), radii=[float(h.r_min), float(v.r_min)],
                tpq=[float(h.t), float(h.p), float(h.q), float(v.t), float(v.p), float(v.q)], sampling=float(m.sampling_size), samples=path)


def main():
    rng = np.random.default_rng(75)
    cases = []
    for s, (R, PL) in enumerate(SETS):
        for k in range(10):
            d = R * float([1.0, 2.5, 5.0, 6.9, 7.1, 9.0, 15.0, 30.0, 45.0, 60.0][k]) * rng.uniform(0.9, 1.1)
            az = rng.uniform(0, 2 * math.pi)
            kind = k % 5
            if kind == 0:      # level
                dz = 0.0
            elif kind == 1:    # gentle climb / descent inside the limits
                dz = d * math.tan(rng.uniform(0.3, 0.8) * (PL[1] if rng.random() < 0.5 else PL[0]))
            elif kind == 2:    # steeper than the limit: spirals
                dz = d * math.tan(min(1.4, 1.5 * PL[1])) * (1 if rng.random() < 0.6 else -1)
            elif kind == 3:    # straight above (take-off): the horizontal problem degenerates
                dz, d = d, 0.0
            else:              # mid-flight: anything
                dz = rng.uniform(-0.5, 0.5) * d
            p0 = rng.uniform(-20, 20, 3) + np.array([0, 0, 60.0])
            p1 = p0 + np.array([d * math.cos(az), d * math.sin(az), dz])
            yaw0 = az + (rng.uniform(-2.5, 2.5) if kind == 4 else rng.uniform(-0.3, 0.3))
            qi = list(p0) + [yaw0 % (2 * math.pi), float(np.clip(rng.uniform(-0.4, 0.4), PL[0], PL[1])) if kind == 4 else 0.0]
            qf = list(p1) + [rng.uniform(0, 2 * math.pi), 0.0 if kind != 4 else float(np.clip(rng.uniform(-0.3, 0.3), PL[0], PL[1]))]
            cases.append((s, [float(x) for x in qi], [float(x) for x in qf]))
    with Pool(int(os.environ.get('KAT_WORKERS', '6'))) as pool:
        res = pool.map(plan, cases, chunksize=1)
    K = len(res)
    off = np.zeros(K + 1, np.int64)
    for k, r in enumerate(res):
        off[k + 1] = off[k] + len(r['samples'])
    out = dict(set=np.array([r['set'] for r in res], np.int32), set_rmin=np.array([s[0] for s in SETS]), set_pitchlims=np.array([s[1] for s in SETS]),
               qi=np.array([r['qi'] for r in res]), qf=np.array([r['qf'] for r in res]), length=np.array([r['length'] for r in res]),
               mode=np.array([r['mode'] for r in res], dtype='S6'), n=np.array([r['n'] for r in res], np.int32),
               radii=np.array([r['radii'] for r in res]), tpq=np.array([r['tpq'] for r in res]), sampling=np.array([r['sampling'] for r in res]),
               samples_off=off, samples=np.concatenate([r['samples'] for r in res]), numpy_version=np.array(np.__version__))
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'F7c_dubins_kat_params.npz'), **out)
    for s, (R, PL) in enumerate(SETS):
        rs = [r for r in res if r['set'] == s]
        print(f'Rmin {R} pitch {PL[0]:.3f} {PL[1]:.3f}:', len(rs), 'plans, length', round(min(r["length"] for r in rs), 2), '..',
              round(max(r["length"] for r in rs), 2), 'radii up to', round(max(max(r["radii"]) for r in rs), 2), 'words', sorted(set(r['mode'] for r in rs)))


if __name__ == '__main__':
    main()
