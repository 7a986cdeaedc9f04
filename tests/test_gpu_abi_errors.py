"""The C-ABI's error behaviour on a live context (-m gpu; tests/test_abi.py checks what can be checked without a GPU): wrong arguments and
calls out of order come back as SCA_ERR_ARG / SCA_ERR_STATE with a message in sca_last_error, never as a crash or a silent wrong answer, and
the context keeps working afterwards.  (The reference raises Python exceptions at the same places: an empty agent list has no kd-tree,
kdTree.py:56-59; a policy is asked for an action only after set_agents, mampenv.py:16-20.)"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OK, ERR_ARG, ERR_STATE, ERR_UNSUPPORTED = 0, -1, -3, -5


@pytest.fixture(scope='module')
def L():
    from sca_amd import _lib
    return _lib.lib()


def _d(a):
    a = np.ascontiguousarray(a, np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def test_create_refuses_bad_sizes_and_null_out(L):
    ctx = C.c_void_p()
    assert L.sca_create(None, 0, 0, 0, C.byref(ctx)) == ERR_ARG
    assert L.sca_create(None, 0, -5, 0, C.byref(ctx)) == ERR_ARG
    assert L.sca_create(None, 0, 10, -1, C.byref(ctx)) == ERR_ARG
    assert L.sca_create(None, 0, 10, 0, None) == ERR_ARG
    assert L.sca_create(None, 0, 10, 0, C.byref(ctx)) == OK and ctx.value
    L.sca_destroy(ctx)


def test_calls_out_of_order_and_bad_arguments_are_refused_and_the_context_survives(L):
    from sca_amd import _lib
    n = 40
    ctx = C.c_void_p()
    assert L.sca_create(None, 0, n, 2, C.byref(ctx)) == OK
    rng = np.random.default_rng(3)
    pos, ppos = _d(rng.uniform(-10, 10, (n, 3)) + [0, 0, 20])
    goal, pgoal = _d(-pos + [0, 0, 40])
    head, phead = _d(np.zeros((n, 3)))
    rad, prad = _d(np.full(n, 0.5))
    ps, pps = _d(np.ones(n))
    mrd, pmrd = _d(np.full(n, 1e9))
    vel = np.zeros((n, 3), np.float32); pvel = vel.ctypes.data_as(C.POINTER(C.c_float))
    flags = np.zeros(n, np.uint8); pflags = flags.ctypes.data_as(C.POINTER(C.c_uint8))
    pol = np.zeros(n, np.uint8); ppol = pol.ctypes.data_as(C.POINTER(C.c_uint8))
    act = np.zeros((n, 7), np.float32); pact = act.ctypes.data_as(C.POINTER(C.c_float))
    err = lambda: L.sca_last_error(ctx).decode()
    # nothing is set yet
    assert L.sca_set_state(ctx, ppos, pvel, phead, pflags, None, None) == ERR_STATE and 'sca_set_agents' in err()
    assert L.sca_policy_pass(ctx, 0) == ERR_STATE
    assert L.sca_run_steps(ctx, 1, 0) == ERR_STATE
    assert L.sca_get_state(ctx, ppos, pvel, phead, pflags, None, None) == ERR_STATE
    # agent sets that cannot be
    assert L.sca_set_agents(ctx, 0, prad, pps, pgoal, ppol, None, pmrd) == ERR_ARG
    assert L.sca_set_agents(ctx, n + 1, prad, pps, pgoal, ppol, None, pmrd) == ERR_ARG
    assert L.sca_set_agents(ctx, n, None, pps, pgoal, ppol, None, pmrd) == ERR_ARG
    bad_pol = pol.copy(); bad_pol[7] = 6
    assert L.sca_set_agents(ctx, n, prad, pps, pgoal, bad_pol.ctypes.data_as(C.POINTER(C.c_uint8)), None, pmrd) == ERR_ARG
    assert L.sca_set_obstacles(ctx, 3, None, None) == ERR_ARG              # more obstacles than the context was created for / no arrays
    # a good set, then a pass without a state
    assert L.sca_set_agents(ctx, n, prad, pps, pgoal, ppol, None, pmrd) == OK
    assert L.sca_policy_pass(ctx, 0) == ERR_STATE
    assert L.sca_set_state(ctx, None, pvel, phead, pflags, None, None) == ERR_ARG
    assert L.sca_set_state(ctx, ppos, pvel, phead, pflags, None, None) == OK
    # neighbour modes that do not exist, negative step counts
    assert L.sca_policy_pass(ctx, 7) == ERR_UNSUPPORTED and 'mode' in err().lower()
    assert L.sca_run_steps(ctx, -1, 0) == ERR_ARG
    assert L.sca_run_steps(ctx, 1, 9) == ERR_UNSUPPORTED
    assert L.sca_get_actions(ctx, None) == ERR_ARG
    # per-agent attributes out of range name the agent
    nd = np.full(n, 10.0); nd[3] = -1.0
    assert L.sca_set_agent_params(ctx, n, nd.ctypes.data_as(C.POINTER(C.c_double)), None, None, None, None, None, None) == ERR_ARG
    assert 'agent 3' in err()
    # ... and after all that the context still answers, with the answer of a context that never saw an error
    assert L.sca_policy_pass(ctx, 0) == OK and L.sca_get_actions(ctx, pact) == OK
    ctx2 = C.c_void_p()
    assert L.sca_create(None, 0, n, 2, C.byref(ctx2)) == OK
    assert L.sca_set_agents(ctx2, n, prad, pps, pgoal, ppol, None, pmrd) == OK
    assert L.sca_set_state(ctx2, ppos, pvel, phead, pflags, None, None) == OK
    act2 = np.zeros((n, 7), np.float32)
    assert L.sca_policy_pass(ctx2, 0) == OK and L.sca_get_actions(ctx2, act2.ctypes.data_as(C.POINTER(C.c_float))) == OK
    assert np.array_equal(act, act2) and act.any()
    L.sca_destroy(ctx)
    L.sca_destroy(ctx2)


def test_contexts_give_their_memory_back():
    """A long-lived service creates and destroys contexts: after 150 contexts that each ran kd, grid and AUTO passes with the device tracker,
    per-agent attributes and the episode log, the device has (within allocator granularity) the free memory it started with."""
    from sca_amd import scenarios, solver as S
    hip = C.CDLL('libamdhip64.so.7')                     # by SONAME: the runtime this process already has (the library's; never a second one)

    def free_bytes():
        free, total = C.c_size_t(0), C.c_size_t(0)
        assert hip.hipDeviceSynchronize() == 0 and hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value

    sc = scenarios.circle(3000)
    n = 3000
    zaxis = S.zaxis_flags(sc['start'], sc['goal'])
    mrd = scenarios.max_run_dist(sc['start'], sc['goal'])

    def once():
        sol = S.BatchedSolver(max_agents=n, max_obstacles=4)
        sol.set_obstacles(np.array([[0.0, 0.0, 3.0]]), np.array([1.0]))
        sol.set_agents(np.full(n, 0.5), np.ones(n), sc['goal'][:, :3], np.where(np.arange(n) % 3 == 0, 3, 0).astype(np.uint8), zaxis, mrd)
        sol.set_agent_params(neighbor_dist=np.where(np.arange(n) % 2 == 0, 8.0, 10.0))
        sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
        sol.device_tracker_enable(sc['goal'][:, 3:6])
        sol.history_enable(16)
        for mode in (S.NBR_KDTREE, S.NBR_AUTO, S.NBR_GRID):
            sol.run_steps(2, mode)
        sol.synchronize()
        sol.close()

    for _ in range(3):
        once()                                               # (first uses: code objects, streams, the allocator's pools)
    free0 = free_bytes()
    for _ in range(150):
        once()
    free1 = free_bytes()
    assert free0 - free1 < 64 << 20, (free0, free1)          # 150 leaked contexts of this size would be gigabytes


def test_two_contexts_stepped_from_two_host_threads_at_once_equal_the_serial_runs():
    """Contexts share nothing but the code object and its constant tables: two episodes stepped concurrently from two host threads (ctypes
    releases the GIL inside every library call), one with the device tracker in kd mode and one without in SCA_NBR_AUTO, walk through the
    states of the same episodes run one after the other."""
    import threading
    from sca_amd import scenarios, solver as S

    def episode(kind, out):
        if kind == 'sca':
            sc, n, pol, mode = scenarios.circle(1500), 1500, np.zeros(1500, np.uint8), S.NBR_KDTREE
        else:
            sc, n, pol, mode = scenarios.random_cube(4096, seed=3), 4096, np.full(4096, 3, np.uint8), S.NBR_AUTO
        sol = S.BatchedSolver(max_agents=n, max_obstacles=1)
        sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
        sol.set_agents(np.full(n, 0.5), np.ones(n), sc['goal'][:, :3], pol, S.zaxis_flags(sc['start'], sc['goal']), scenarios.max_run_dist(sc['start'], sc['goal']))
        sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
        if kind == 'sca':
            sol.device_tracker_enable(sc['goal'][:, 3:6])
        snaps = []
        for t in range(30):
            sol.run_steps(3, mode)
            sol.synchronize()
            g = sol.get_state()
            snaps.append((g['pos'], g['vel'], g['flags'], sol.get_kd_perm()))
        sol.close()
        out[kind] = snaps

    serial, both = {}, {}
    episode('sca', serial)
    episode('orca', serial)
    th = [threading.Thread(target=episode, args=(k, both)) for k in ('sca', 'orca')]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in ('sca', 'orca'):
        assert len(both[k]) == 30
        for i, (a, b) in enumerate(zip(serial[k], both[k])):
            for x, y in zip(a, b):
                assert np.array_equal(x, y), (k, i)


def test_torch_imported_after_the_library_still_sees_the_gpu():
    """One HIP runtime per process (sca_amd/_lib.py::_share_torch_hip_runtime): PyTorch-ROCm ships its own libamdhip64 under the system's
    SONAME, and with libsca_hip.so loaded first on the system's runtime a later `import torch` found no GPU.  A fresh process that steps a
    swarm FIRST and imports torch AFTERWARDS must see the device, share a buffer with the library, and get the same velocities as a
    process that never imports torch."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from sca_amd import scenarios, solver as S
sc = scenarios.circle(500)
sol = S.BatchedSolver(max_agents=500)
sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
sol.set_agents(np.full(500, 0.5), np.ones(500), sc['goal'][:, :3], np.full(500, 3, np.uint8), S.zaxis_flags(sc['start'], sc['goal']), scenarios.max_run_dist(sc['start'], sc['goal']))
sol.set_state(sc['start'][:, :3], np.zeros((500, 3), np.float32), sc['start'][:, 3:6], np.zeros(500, np.uint8))
sol.run_steps(5, 0); sol.synchronize()
if WITH_TORCH:
    import torch
    assert torch.cuda.is_available(), 'torch sees no GPU after libsca_hip.so'
    t = torch.arange(8, device='cuda', dtype=torch.float64)
    assert float((t * 2).sum().item()) == 56.0
sol.run_steps(5, 0); sol.synchronize()
print('VEL', sol.get_state()['vel'].tobytes().hex()[:4000])
''' % root
    outs = []
    for with_torch in (True, False):
        r = subprocess.run([sys.executable, '-c', code.replace('WITH_TORCH', str(with_torch))], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith('VEL')][0])
    assert outs[0] == outs[1]
