"""The cell-owner partition of SCA_NBR_GRID with halo exchange (sca_partition_*, sca_amd/csrc/sca_partition.hip.h; SURVEY.md
8(f)-4, the half round 2 left open): ranks that hold only their slab of grid cells plus a one-cell halo, exchanging with their
two slab neighbours the records next to the cut and the agents that crossed it, must reproduce the single-rank grid run bit for
bit -- neighbour lists are ordered by (distSq, obstacle first, id), i.e. independent of who holds what; every collision pair is
evaluated by the owner of either side from the same records; an agent migrates with its heading, distances, v_pref and tracker
record.  Two / three ranks on ONE GPU (gloo, host-staged transfers; the GPU runs use the same calls over RCCL)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene(n, kind):
    from sca_amd import scenarios, solver as S
    if kind == 'cube':                                          # every policy, dense enough for collisions and arrivals
        sc = scenarios.random_cube(n, seed=3)
        pol = (np.arange(n) % 5).astype(np.uint8)
    elif kind == 'arrive':                                      # short trips: most agents reach their goal within 60 steps, many of them
        sc = scenarios.random_cube(n, seed=11)                  # next to a cut -- as halo copies of the rank across it (the
        rng = np.random.default_rng(12)                         # at-goal flag of a halo copy must be set in the step it arrives)
        sc['goal'] = sc['goal'].copy()
        sc['goal'][:, :3] = sc['start'][:, :3] + rng.normal(0.0, 1.6, (n, 3))
        pol = np.where(np.arange(n) % 3 == 0, 1, np.arange(n) % 5).astype(np.uint8)      # RVO3D / S-RVO3D / ORCA: straight-line v_pref
        pol[pol == 0] = 2
    else:
        sc = scenarios.circle(n)
        pol = np.where(np.arange(n) % 7 == 3, 2, 0).astype(np.uint8)
    return sc, pol


def _solver(sc, pol, n, track, hetero=0):
    """hetero: the reference's per-Agent attributes (agent.py:24-41) drawn per agent with the value sets of the F17 / F18 fixtures
    (tools/gen_golden.py::_hetero, _hetero_track); 2: every tracked agent its own (turning_radius, pitchlims) -- the per-agent form."""
    import math
    from sca_amd import scenarios, solver as S
    sol = S.BatchedSolver(max_agents=n, max_obstacles=4)
    sol.set_obstacles(np.array([[0.0, 0.0, 30.0]]), np.array([1.5]))
    sol.set_agents(np.full(n, 0.5), np.full(n, 1.0), sc['goal'][:, :3], pol, S.zaxis_flags(sc['start'], sc['goal']),
                   scenarios.max_run_dist(sc['start'], sc['goal']))
    rng = np.random.default_rng(1700 + hetero)
    if hetero:
        sol.set_agent_params(neighbor_dist=rng.choice([3.0, 6.5, 10.0], n), max_neighbors=rng.choice([2, 5, 9, 16], n).astype(np.int32),
                             time_step=rng.choice([0.05, 0.1, 0.2], n), time_horizon=rng.choice([2.0, 5.0, 10.0], n),
                             max_speed=rng.choice([0.8, 1.0, 1.5], n), max_heading_change=rng.choice([0.5, math.pi / 4, 1.0, math.pi / 2], n),
                             dt_nominal=rng.choice([0.05, 0.1], n))
    sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
    if track:
        sol.device_tracker_enable(sc['goal'][:, 3:6])
        if hetero == 1:
            pl = np.array([(-math.pi / 4, math.pi / 4), (-math.pi / 6, math.pi / 6), (-0.5, 0.9)])[rng.integers(0, 3, n)]
            sol.device_tracker_set_agent_params(rng.choice([0.8, 1.5, 3.0], n), pl[:, 0].copy(), pl[:, 1].copy())
        elif hetero == 2:
            sol.device_tracker_set_agent_params(0.8 + 2.2 * rng.random(n), -(0.35 + 0.4 * rng.random(n)), 0.35 + 0.5 * rng.random(n))
    return sol


@pytest.mark.parametrize('track', [False, True])
def test_one_rank_partition_equals_plain_grid_run(track):
    """The indirection alone (kernels take their agents from the owned list, the grid is built over the present list; the lists are
    rebuilt every step in arbitrary order) changes nothing."""
    from sca_amd import solver as S
    n, steps = 2500, 30
    sc, pol = _scene(n, 'cube')
    a, b = _solver(sc, pol, n, track), _solver(sc, pol, n, track)
    b.partition_init(0, 1, axis=0)
    assert b.partition_counts() == (n, 0) and sorted(b.partition_owned().tolist()) == list(range(n))
    for _ in range(steps):
        a.run_steps(1, S.NBR_GRID); b.run_steps(1, S.NBR_GRID)
    a.synchronize(); b.synchronize()
    sa, sb = a.get_state(), b.get_state()
    for k in ('pos', 'vel', 'heading', 'flags', 'total_dist', 'step_num'):
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(a.actions(), b.actions())
    if track:
        assert np.array_equal(a.device_tracker_replans(), b.device_tracker_replans())
    a.close(); b.close()


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
import numpy as np, torch, torch.distributed as dist
from sca_amd import solver as S
from sca_amd.distributed import PartitionedStepper
from test_gpu_partition import _scene, _solver
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('gloo')
n, steps, kind, track = int(os.environ['SCA_TEST_N']), int(os.environ['SCA_TEST_STEPS']), os.environ['SCA_TEST_SCENE'], bool(int(os.environ['SCA_TEST_TRACK']))
hetero = int(os.environ.get('SCA_TEST_HETERO', '0'))
sc, pol = _scene(n, kind)
sol = _solver(sc, pol, n, track, hetero)
st = PartitionedStepper(sol, rank, world, torch, dist, axis=int(os.environ.get('SCA_TEST_AXIS', '0')), staged=True)
own0 = set(st.owned().tolist())
ref = _solver(sc, pol, n, track, hetero)
moved_total = 0
ok = True
for block in range(steps // 10):
    st.run(10); st.sync()
    ref.run_steps(10, S.NBR_GRID); ref.synchronize()
    own = st.owned()
    got, want = sol.get_state(), ref.get_state()
    for k in ('pos', 'vel', 'heading', 'flags', 'total_dist', 'step_num'):
        if not np.array_equal(got[k][own], want[k][own]):
            bad = np.flatnonzero((got[k][own] != want[k][own]).reshape(len(own), -1).any(1))
            print('RANK', rank, 'block', block, 'DIFF', k, len(bad), 'of', len(own), 'first', own[bad[:6]].tolist(), flush=True)
            ok = False
    if not np.array_equal(sol.actions()[own], ref.actions()[own]):
        print('RANK', rank, 'block', block, 'DIFF action', flush=True); ok = False
    if track and not np.array_equal(sol.device_tracker_replans()[own], ref.device_tracker_replans()[own]):
        print('RANK', rank, 'block', block, 'DIFF replans', flush=True); ok = False
    # the ranks' owned sets partition the swarm
    cnt = torch.zeros(n, dtype=torch.int32); cnt[torch.from_numpy(own.astype(np.int64))] = 1
    dist.all_reduce(cnt)
    if not bool((cnt == 1).all()):
        print('RANK', rank, 'block', block, 'ownership is not a partition:', int((cnt != 1).sum()), flush=True); ok = False
    moved_total = len(own0 ^ set(own.tolist()))
t = torch.tensor([moved_total]); dist.all_reduce(t)
print('RANK', rank, 'owned', len(st.owned()), 'halo', sol.partition_counts()[1], 'changed owner (both ranks)', int(t.item()), 'OK' if ok else 'MISMATCH', flush=True)
if kind == 'arrive':
    arrived = int((ref.get_state()['flags'] & 1).sum())
    print('RANK', rank, 'arrived', arrived, flush=True)
    if arrived < n // 4:
        print('RANK', rank, 'too few arrivals: the test does not test the at-goal flag of halo copies', flush=True); ok = False
if int(os.environ.get('SCA_TEST_NEED_MIGRATION', '1')) and int(t.item()) == 0:
    print('RANK', rank, 'no agent crossed a cut: the test does not test migration', flush=True); ok = False
dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


@pytest.mark.parametrize('world,n,kind,track,axis,steps,hetero', [(2, 3000, 'cube', False, 0, 60, 0), (2, 3000, 'cube', True, 1, 60, 0),
                                                                 (3, 4000, 'cube', True, 2, 40, 0), (2, 20000, 'circle', True, 0, 30, 0),
                                                                 (2, 6000, 'arrive', False, 0, 60, 0), (3, 6000, 'arrive', False, 1, 60, 0),
                                                                 # round 6: per-agent attributes under the partition (round 5 refused them)
                                                                 (2, 3000, 'cube', True, 0, 60, 1), (3, 4000, 'cube', True, 1, 40, 1),
                                                                 (2, 3000, 'cube', True, 2, 40, 2), (2, 6000, 'arrive', False, 0, 60, 1)])
def test_partitioned_ranks_match_single_rank(tmp_path, world, n, kind, track, axis, steps, hetero):
    """hetero = 1: the solver attributes drawn per agent (sca_set_agent_params, the F17 value sets) and, for the tracked agents, three
    turning radii x three pitch-limit pairs (classes, F18's value sets); 2: every tracked agent its own planner attributes (the per-agent
    form).  They are static per-agent inputs, replicated on every rank by global id like sca_set_agents' -- the slabs are cut in cells of
    the LARGEST neighborDist."""
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    port = str(29560 + world + axis + (10 if kind == 'arrive' else 0) + 20 * hetero)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=port, SCA_TEST_N=str(n), SCA_TEST_STEPS=str(steps), SCA_TEST_SCENE=kind,
               SCA_TEST_HETERO=str(hetero),
               SCA_TEST_TRACK=str(int(track)), SCA_TEST_AXIS=str(axis), SCA_TEST_NEED_MIGRATION='0' if kind in ('circle', 'arrive') else '1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
                        '--master-port', port, str(script), ROOT], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-3000:]
    assert 'MISMATCH' not in r.stdout and 'DIFF' not in r.stdout, r.stdout[-3000:]      # (every rank exits 0 only when it agrees)


def test_partition_and_per_agent_attributes_in_either_order():
    """Round 5 refused the combination.  Now: attributes first, then the partition (cells of the largest neighborDist) -- fine; attributes
    under a standing partition -- fine while the cell size stays what the slabs were cut with, refused (with the reason) when it would change."""
    from sca_amd import scenarios, solver as S
    n = 2000
    sc = scenarios.random_cube(n, seed=3)
    sol = S.BatchedSolver(max_agents=n, max_obstacles=1)
    sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
    sol.set_agents(np.full(n, 0.5), np.ones(n), sc['goal'][:, :3], np.full(n, 3, np.uint8), S.zaxis_flags(sc['start'], sc['goal']),
                   scenarios.max_run_dist(sc['start'], sc['goal']))
    sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
    sol.set_agent_params(neighbor_dist=np.where(np.arange(n) % 2 == 0, 10.0, 6.0))
    sol.partition_init(0, 2, axis=0)                               # largest range 10: the cells of the default
    sol.set_agent_params(neighbor_dist=np.where(np.arange(n) % 3 == 0, 10.0, 4.0), max_neighbors=np.full(n, 8, np.int32))   # still 10
    with pytest.raises(S.ScaError, match='cell size'):
        sol.set_agent_params(neighbor_dist=np.full(n, 8.0))
    sol.set_agent_params()                                         # back to one value per context (10: the same cells)
    sol.close()
