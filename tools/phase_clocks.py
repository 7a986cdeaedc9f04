"""Per-phase clocks of two latency-bound kernels, from DEBUG builds of the library (needs a GPU; the numbers in DESIGN.md sections 3 and 5):

    SCA_BUILD_DEFS=-DSCA_KB_TIMING python -m sca_amd.build && python tools/phase_clocks.py kd [N]        # k_kd_block, per tree level
    SCA_BUILD_DEFS=-DSCA_KT_TIMING python -m sca_amd.build && python tools/phase_clocks.py track c2|c4|c5 # k_track and track_decide
    python -m sca_amd.build                                                                               # back to the product build

One lane (workgroup 0, thread 0) reads the 100-MHz wall clock between the phases; the debug entry points exist in those builds only."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, '.')
from sca_amd import _lib, scenarios, solver as S


def reader(name):
    L = _lib.lib()
    try:
        f = getattr(L, name)
    except AttributeError:
        sys.exit(f'{name} is not in this build of the library: see the docstring')
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
    return f


def kd(n):
    sc = scenarios.random_cube(n, seed=0)
    sol = S.BatchedSolver(max_agents=n, max_obstacles=1)
    sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
    sol.set_agents(np.full(n, 0.5), np.ones(n), sc['goal'][:, :3], np.full(n, 3, np.uint8), S.zaxis_flags(sc['start'], sc['goal']),
                   scenarios.max_run_dist(sc['start'], sc['goal']))
    sol.set_state(sc['start'][:, :3], np.full((n, 3), 0.3, np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
    read = reader('sca_debug_read_ps')
    for _ in range(3):
        sol.policy_pass(S.NBR_KDTREE)
    out = np.zeros(200, np.int32)
    read(sol.ctx, out.ctypes.data_as(C.POINTER(C.c_int)), 200)
    print('k_kd_block, ticks of 10 ns per level: A boxes, B split + scan, B2 wavefront totals, C ranks, D swaps + records')
    for r in out[:60].reshape(-1, 5)[:10]:
        print(r.tolist(), 'sum', int(r.sum()))


def track(kind):
    if kind == 'c5':
        sc = scenarios.takeoff_landing(16384)
        n = len(sc['start'])
        policy = np.where(np.arange(n) % 2 == 0, 0, 2).astype(np.uint8)
    else:
        n = 100000 if kind == 'c4' else 1024
        sc = scenarios.circle(n)
        policy = np.zeros(n, np.uint8)
    sol = S.BatchedSolver(max_agents=n, max_obstacles=max(1, len(sc['obs_radius'])))
    sol.set_obstacles(sc['obs_pos'], sc['obs_radius'])
    sol.set_agents(np.full(n, 0.5), np.ones(n), sc['goal'][:, :3], policy, S.zaxis_flags(sc['start'], sc['goal']),
                   scenarios.max_run_dist(sc['start'], sc['goal']))
    sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
    sol.policy_pass(S.NBR_AUTO)                      # (allocates the list the clocks are written to)
    sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
    sol.device_tracker_enable(sc['goal'][:, 3:6], in_pass=True)
    read = reader('sca_debug_read_kdq')
    for _ in range(3):
        sol.run_steps(20)
        sol.synchronize()
        out = np.zeros(48, np.int32)
        read(sol.ctx, out.ctypes.data_as(C.POINTER(C.c_int)), 48)
        print(kind, 'k_track, ticks of 10 ns: tables into LDS, barrier, loads, track_decide, finish / bucket, barrier, list:', out[:7].tolist(),
              'sum', int(out[:7].sum()))
        print('    track_decide: norm to the goal, update_dubins, norm + cosine, acos, is_parallel, update_dubins again:', out[17:23].tolist())
        g = out[32:48].tolist()
        print(f'    the speculative search of one plan ({g[15]} rounds behind the first): frames {g[0]}, first round: path {g[1]} + candidate {g[2]} + walk {g[3]}; '
              f'rounds: tree + path {g[7] + g[4]}, candidate {g[5]}, walk {g[6]}; the winner\'s maneuvers {g[8]}, finish_plan {g[9]}, adopt + v_pref + prologue {g[10]}')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'kd':
        kd(int(sys.argv[2]) if len(sys.argv) > 2 else 4096)
    elif len(sys.argv) > 2 and sys.argv[1] == 'track':
        track(sys.argv[2])
    else:
        sys.exit(__doc__)
