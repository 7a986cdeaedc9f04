"""Two ranks on ONE GPU (gloo, host-staged exchange): the real libsca_hip step_begin / exchange / step_end path with
sharded agents must reproduce the single-rank resident run bit for bit.  (The 8-GPU runs use RCCL on the same code.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from sca_amd import scenarios, solver as S
from sca_amd.distributed import ShardedStepper
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('gloo')
n, steps = int(os.environ.get('SCA_TEST_N', '3000')), 12
if n > 50000 and os.environ.get('SCA_TEST_SCENE') == 'mixed':  # every policy in every shard (LP agents through k_lp), random cube
    sc = scenarios.random_cube(n, seed=4)
    pol = (np.arange(n) % 5).astype(np.uint8)
elif n > 50000:                                               # a swarm whose shards get the split solve and the fused tracker kernel
    sc = scenarios.circle(n)
    pol = np.where(np.arange(n) % 11 == 5, 2, 0).astype(np.uint8)
else:
    sc = scenarios.random_cube(n, seed=2)
    pol = np.where(np.arange(n) % 3 == 0, 0, np.where(np.arange(n) % 3 == 1, 3, 4)).astype(np.uint8)

def make():
    sol = S.BatchedSolver(max_agents=n, max_obstacles=1, device=0)
    sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
    sol.set_agents(np.full(n, 0.5), np.full(n, 1.0), sc['goal'][:, :3], pol, S.zaxis_flags(sc['start'], sc['goal']),
                   scenarios.max_run_dist(sc['start'], sc['goal']))
    sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
    if os.environ.get('SCA_TEST_HETERO'):                      # every agent its own solver attributes (sca_set_agent_params), the same arrays on every rank
        rng = np.random.default_rng(9)
        sol.set_agent_params(neighbor_dist=rng.choice([4.0, 10.0, 15.0], n), max_neighbors=rng.choice([4, 9, 16], n), time_horizon=rng.choice([3.0, 10.0], n),
                             max_speed=rng.choice([1.0, 1.5], n), max_heading_change=rng.choice([0.6, 0.7853981633974483], n), time_step=rng.choice([0.1, 0.2], n))
    if os.environ.get('SCA_TEST_TRACK'):
        sol.device_tracker_enable(sc['goal'][:, 3:6])          # the SCA third of the agents follows Dubins paths, per shard
        if os.environ.get('SCA_TEST_HETERO'):
            rng = np.random.default_rng(10)
            sol.device_tracker_set_agent_params(turning_radius=rng.choice([1.0, 1.5, 2.5], n))
    return sol

mode = int(os.environ.get('SCA_TEST_MODE', '0'))             # 0 kd-tree, 1 grid
sol = make()
st = ShardedStepper(sol, rank, world, torch_mod=torch, dist_mod=dist, staged=True, mode=mode)
st.run(steps); st.sync()
got = sol.get_state()
ref_sol = make()
ref_sol.run_steps(steps, 0 if mode == 3 else mode); ref_sol.synchronize()      # (SCA_NBR_AUTO must equal the kd-tree run)
ref = ref_sol.get_state()
lo, hi = st.begin, st.begin + st.count
ok = np.array_equal(got['pos'], ref['pos']) and np.array_equal(got['vel'], ref['vel'])
ok = ok and np.array_equal(got['flags'][lo:hi], ref['flags'][lo:hi]) and np.array_equal(got['flags'] & 1, ref['flags'] & 1)
ok = ok and np.array_equal(got['heading'][lo:hi], ref['heading'][lo:hi]) and np.array_equal(got['total_dist'][lo:hi], ref['total_dist'][lo:hi])
ok = ok and (mode not in (0, 3) or np.array_equal(sol.get_kd_perm(), ref_sol.get_kd_perm()))
state_ok = ok
if not ok:                                                    # say what differs (a bug report's worth)
    for k in ('pos', 'vel', 'heading', 'flags', 'total_dist'):
        dif = np.flatnonzero((got[k] != ref[k]).reshape(n, -1).any(1))
        if dif.size:
            print('RANK', rank, 'DIFF', k, dif.size, 'agents, first', dif[:8].tolist(), 'last', int(dif[-1]), 'shard', (lo, hi),
                  'max', float(np.abs(got[k].astype(np.float64) - ref[k].astype(np.float64)).max()), flush=True)
if n > 50000 and os.environ.get('SCA_TEST_TRACK'):
    f = sol.pass_forms()
    # the split is decided from the shard size; k_track_replan needs a re-plan count to have come back (asynchronous): the fused
    # form against the plain one is tests/test_gpu_tracker.py's business, here it may or may not have been reached yet
    ok = ok and (bool(f & S.FORM_SOLVE_SPLIT) or os.environ.get('SCA_SOLVE_SPLIT') == '0' or os.environ.get('SCA_TEST_SCENE') == 'mixed')
    print('RANK', rank, 'forms', f, 'state', state_ok, flush=True)
print('RANK', rank, 'OK' if ok else 'MISMATCH', flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


@pytest.mark.parametrize('track,mode,n,world', [(False, 0, 3000, 2), (True, 0, 3000, 2), (False, 1, 3000, 2), (True, 1, 3000, 2), (True, 0, 120000, 2),
                                                (True, 0, -60000, 2), (False, 3, 3000, 2), (False, 3, -60000, 2), (True, 3, 3000, 2),
                                                # 1 / 2 / 4 / 8 shards bit-identical to one -- kd-tree, grid, with and without the tracker
                                                (True, 0, 4096, 4), (True, 1, 4096, 4), (False, 3, 4096, 4),
                                                (True, 0, 4096, 8), (True, 1, 4096, 8), (False, 0, 4096, 8), (False, 3, 4096, 8),
                                                # every agent its own solver / planner attributes, the same arrays on every rank (n encodes it: + 1)
                                                (True, 0, 3001, 2), (False, 3, 3001, 2), (True, 1, 3001, 3)])
def test_two_ranks_one_gpu_match_single_rank(tmp_path, track, mode, n, world):
    """track=True: with the device-side v_pref tracker inside every step (tracker records are shard-local, its re-plans run
    next to the replicated kd build).  mode 1: SCA_NBR_GRID (the grid is replicated, the queries sharded).  mode 3: SCA_NBR_AUTO on two
    ranks against the single-rank KD-TREE run.  n = 120 000: shards
    of 60 000 agents, which get k_track_replan and the split solve, the second one with shard_begin != 0 -- the case that found
    the missing fence of the one-launch-per-level kd build (two processes on one GPU is also a scheduling stress).  n < 0: |n|
    agents of all five policies in a random cube.  world = 4 / 8: as many processes sharing GPU 0, shards of 1024 / 512 agents."""
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    hetero = abs(n) % 1000 == 1
    if hetero:
        n -= 1
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', SCA_TEST_MODE=str(mode), SCA_TEST_N=str(abs(n)))
    if hetero:
        env['SCA_TEST_HETERO'] = '1'
    if n < 0:
        env['SCA_TEST_SCENE'] = 'mixed'                         # every policy in every shard, random cube
    if track:
        env['SCA_TEST_TRACK'] = '1'
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
                        '--master-addr', '127.0.0.1', '--master-port', '29541', str(script), ROOT],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count('OK') == world, r.stdout[-3000:]


RCCL_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from sca_amd import scenarios, solver as S
from sca_amd.distributed import ShardedStepper
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))          # backend "nccl" is RCCL on ROCm
n, steps = 4096, 10
sc = scenarios.circle(n)

def make():
    sol = S.BatchedSolver(max_agents=n, max_obstacles=1, device=0)
    sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
    sol.set_agents(np.full(n, 0.5), np.full(n, 1.0), sc['goal'][:, :3], np.zeros(n, np.uint8), S.zaxis_flags(sc['start'], sc['goal']),
                   scenarios.max_run_dist(sc['start'], sc['goal']))
    sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
    if os.environ.get('SCA_TEST_TRACK'):
        sol.device_tracker_enable(sc['goal'][:, 3:6])
    return sol

sol = make()
st = ShardedStepper(sol, 0, 1, torch_mod=torch, dist_mod=dist, force_exchange=True)   # device buffers straight into the collective
st.run(steps); st.sync()
dist.barrier(); torch.cuda.synchronize()
got = sol.get_state()
ref_sol = make()
ref_sol.run_steps(steps); ref_sol.synchronize()
ref = ref_sol.get_state()
ok = all(np.array_equal(got[k], ref[k]) for k in ('pos', 'vel', 'heading', 'flags', 'total_dist', 'step_num'))
print('RCCL', 'OK' if ok else 'MISMATCH', flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


@pytest.mark.parametrize('track', [False, True])
def test_rccl_in_place_all_gather_path_single_rank(tmp_path, track):
    """The exchange exactly as bench.py --gpus N runs it (bound device record buffers, in-place all_gather_into_tensor on
    the RCCL backend, library kernels on torch's current stream), with the one rank a 1-GPU box offers.  track=True adds the
    device-side v_pref tracker (its side stream forks from and joins torch's stream)."""
    script = tmp_path / 'rccl_worker.py'
    script.write_text(RCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29543', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    if track:
        env['SCA_TEST_TRACK'] = '1'
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'RCCL OK' in r.stdout, r.stdout[-3000:]


def test_bench_script_two_rank_path():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per process), with the ranks sharing
    GPU 0 through the script's test hook: one JSON line from rank 0, whole-job agent-steps summed over the ranks, strong
    scaling by default (the workload's agent count in total)."""
    import json
    env = dict(os.environ, SCA_BENCH_SHARE_GPU='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29547')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', '29547', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6',
                        '--warmup', '3', '--agents', '6000'], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['scaling'] == 'strong' and out['config']['agents'] == 6000
    assert out['config']['agents_per_gpu'] == 3000 and out['config']['agent_steps_timed'] == 6000 * 6
    assert out['config']['re_plans_timed'] > 0                      # SCA as shipped: the tracker runs inside the step
    assert out['value'] > 0 and 'roofline' in out and 'cpu_baseline' not in out and 'scale_model' not in out


ORDER_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from sca_amd import scenarios, solver as S
from sca_amd.distributed import ShardedStepper
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
n = 60000
sc = scenarios.circle(n)
sol = S.BatchedSolver(max_agents=n, max_obstacles=1, device=0)
sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
sol.set_agents(np.full(n, 0.5), np.ones(n), sc['goal'][:, :3], np.zeros(n, np.uint8), S.zaxis_flags(sc['start'], sc['goal']),
               scenarios.max_run_dist(sc['start'], sc['goal']))
sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
st = ShardedStepper(sol, 0, 1, torch_mod=torch, dist_mod=dist, force_exchange=True)
assert st._stream.cuda_stream != 0                  # a stream the stepper owns
st.run(3); st.sync(); torch.cuda.synchronize()
stale = 0
for it in range(10):
    st.run(4)                                       # a backlog of queued work on the stepper's stream
    sol.step_begin(0)
    full, mine = st._moved_records()
    tmp = torch.zeros_like(mine)
    with torch.cuda.stream(st._stream):
        dist.all_gather_into_tensor(tmp, mine)      # out of place: a gather that ran too early would copy old records
    sol.step_end()
    sol.synchronize(); torch.cuda.synchronize()
    a = tmp.view(-1, 48)[:, :24].cpu().numpy(); b = mine.view(-1, 48)[:, :24].cpu().numpy()   # positions: step_end leaves them alone
    stale += int((a != b).any())
print('ORDER', 'OK' if stale == 0 else 'STALE %d' % stale, flush=True)
dist.destroy_process_group()
sys.exit(0 if stale == 0 else 1)
'''


def test_collective_is_ordered_behind_the_library_kernels(tmp_path):
    """The library launches on the stepper's own torch stream and the RCCL collective is issued with that stream current, so
    an all-gather behind sca_step_begin sees the records that call wrote even when the stream holds a backlog.  (The
    library's default is a non-blocking stream of its own, which nothing in torch orders against.)"""
    script = tmp_path / 'order_worker.py'
    script.write_text(ORDER_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29553', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'ORDER OK' in r.stdout, r.stdout[-3000:]


INLIB_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from sca_amd import scenarios, solver as S
from sca_amd.distributed import ShardedStepper
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
local = int(os.environ.get('LOCAL_RANK', '0'))
torch.cuda.set_device(local)
dist.init_process_group('gloo')                        # only carries the 128-byte ncclUniqueId; the data path is the library's RCCL
mode = int(os.environ.get('SCA_TEST_MODE', '0'))
n, steps = 4096, 12
sc = scenarios.random_cube(n, seed=4)
pol = np.where(np.arange(n) % 2 == 0, 0, 3).astype(np.uint8)

def make(dev):
    sol = S.BatchedSolver(max_agents=n, max_obstacles=1, device=dev)
    sol.set_obstacles(np.zeros((0, 3)), np.zeros(0))
    sol.set_agents(np.full(n, 0.5), np.full(n, 1.0), sc['goal'][:, :3], pol, S.zaxis_flags(sc['start'], sc['goal']),
                   scenarios.max_run_dist(sc['start'], sc['goal']))
    sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
    if os.environ.get('SCA_TEST_TRACK'):
        sol.device_tracker_enable(sc['goal'][:, 3:6])
    return sol

sol = make(local)
box = [sol.comm_unique_id() if rank == 0 else None]
dist.broadcast_object_list(box, src=0)
st = ShardedStepper(sol, rank, world, mode=mode, inlib=True, unique_id=box[0], force_exchange=True)
st.run(steps); st.sync()                              # ONE sca_run_steps call: 12 x (shard pass, ncclAllGather, flags)
got = sol.get_state()
ref_sol = make(local)
ref_sol.run_steps(steps, mode); ref_sol.synchronize()
ref = ref_sol.get_state()
lo, hi = st.begin, st.begin + st.count
ok = np.array_equal(got['pos'], ref['pos']) and np.array_equal(got['vel'], ref['vel'])
ok = ok and np.array_equal(got['flags'][lo:hi], ref['flags'][lo:hi]) and np.array_equal(got['flags'] & 1, ref['flags'] & 1)
ok = ok and np.array_equal(got['heading'][lo:hi], ref['heading'][lo:hi]) and np.array_equal(got['total_dist'][lo:hi], ref['total_dist'][lo:hi])
print('INLIB', rank, 'OK' if ok else 'MISMATCH', flush=True)
sol.comm_destroy()
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
"""


def _gpu_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize('track,mode', [(False, 0), (True, 0), (False, 1)])
def test_rccl_inside_the_library_single_rank(tmp_path, track, mode):
    """sca_comm_init + sca_run_steps: the library's own ncclAllGather (in place, on its stream, between its kernels) with the
    one rank a 1-GPU box offers must leave the single-GPU run untouched."""
    script = tmp_path / 'inlib_worker.py'
    script.write_text(INLIB_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29561', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY='0', SCA_TEST_MODE=str(mode))
    if track:
        env['SCA_TEST_TRACK'] = '1'
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'INLIB 0 OK' in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize('mode', [0, 1])
def test_rccl_inside_the_library_two_ranks_bit_identical(tmp_path, mode):
    """Two ranks on two GPUs over RCCL (no gloo on the data path): every rank's copy of the swarm equals the single-GPU run
    bit for bit.  Needs two GPUs: skipped on the 1-GPU boxes (RCCL refuses two ranks on one device)."""
    if _gpu_count() < 2:
        pytest.skip('needs 2 GPUs: RCCL does not run two ranks on one device')
    script = tmp_path / 'inlib_worker.py'
    script.write_text(INLIB_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29563', HSA_ENABLE_IPC_MODE_LEGACY='0', SCA_TEST_MODE=str(mode))
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', '29563', str(script), ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count('OK') == 2, r.stdout[-3000:]
