"""CPU tests of the host-side logic: scenario generators against the reference's own start states (golden fixtures),
the drop-in API surface, and the 2-rank sharded stepping protocol (gloo) with a checker backend built on the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from golden_util import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_circle_generator_matches_reference_start_state():
    from sca_amd import scenarios
    fx = load('F1_sca_circle8')                       # run_sca.set_circle_pos((0,0), 10.0, 8)
    sc = scenarios.circle(8, rad=10.0)
    assert np.array_equal(sc['start'], fx['start'])
    assert np.array_equal(sc['goal'], fx['goal6'])
    assert np.array_equal(scenarios.max_run_dist(sc['start'], sc['goal']), fx['max_run_dist'])
    fx = load('F4_sca_circle16_obs')
    sc = scenarios.circle(16, rad=10.0)
    assert np.array_equal(sc['start'], fx['start'])


def test_takeoff_generator_matches_reference_cell():
    from sca_amd import scenarios
    fx = load('F4_sca_takeoff16')                     # run_sca.set_takeoff_landing_pos(16) + build_obstacles
    sc = scenarios.takeoff_landing(16)
    assert np.array_equal(sc['start'], fx['start'])
    assert np.array_equal(sc['goal'], fx['goal6'])
    assert np.array_equal(sc['obs_pos'], fx['obs_pos'])
    assert np.array_equal(sc['obs_radius'], fx['obs_radius'])
    big = scenarios.takeoff_landing(64)
    assert big['start'].shape == (64, 6) and big['obs_pos'].shape == (32, 3)


def test_random_generator_is_seeded_and_separated():
    from sca_amd import scenarios
    a = scenarios.random_cube(300, seed=3)
    b = scenarios.random_cube(300, seed=3)
    assert np.array_equal(a['start'], b['start']) and np.array_equal(a['goal'], b['goal'])
    p = a['start'][:, :3]
    d = np.sqrt(((p[:, None] - p[None]) ** 2).sum(-1)) + np.eye(300) * 1e9
    assert d.min() >= 1.5
    assert a['start'][:, 2].min() >= 5.0 and a['goal'][:, 2].min() >= 5.0        # above the ground plane of util.py:16, like the reference's cube
    assert scenarios.random_cube(4096, seed=0)['start'][:, 2].min() >= 5.0


def test_drop_in_api_surface():
    from sca_amd import env as E
    a = E.Agent(start_pos=[1, 2, 3, 0.5, 0, 0], goal_pos=[4, 6, 3, 0, 0, 0], vel=[0.0, 0.0, 0.0], radius=0.5,
                pref_speed=1.0, policy=E.SCAPolicy, id=0, dt=0.1)
    # agent.py:13-42,70-74
    assert a.policy.type == 'internal' and a.policy.now_goal is None
    assert np.array_equal(a.pos_global_frame, [1, 2, 3]) and np.array_equal(a.heading_global_frame, [0.5, 0, 0])
    assert (a.maxNeighbors, a.neighborDist, a.timeHorizon, a.maxSpeed, a.timeStep) == (16, 10.0, 10.0, 1.0, 0.1)
    assert a.max_run_dist == 15.0 and not a.is_at_goal and not a.is_collision and not a.is_out_of_max_time
    ob = E.Obstacle(pos=[0, 0, 5], shape_dict={'shape': 'cube', 'feature': (2.0, 2.0, 2.0)}, id=0)
    assert abs(ob.radius - 3 ** 0.5) < 1e-15 and ob.is_at_goal           # obstacle.py:9-11,22
    for cls, pid in ((E.SCAPolicy, 0), (E.RVO3DPolicy, 1), (E.SRVO3DPolicy, 2), (E.ORCA3DPolicy, 3),
                     (E.ORCA3DPolicyOfficial, 4), (E.RVO3dDubinsPolicy, 5)):
        assert cls().policy_id == pid
    env = E.MACAEnv()
    with pytest.raises(TypeError):
        env.set_agents([a], obstacles=None)


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
import numpy as np, torch, torch.distributed as dist
from sharded_checker import CheckerBackend, reference_run, scene
from sca_amd.distributed import ShardedStepper
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo')
sc = scene()
be = CheckerBackend(sc)
st = ShardedStepper(be, rank, world, torch_mod=torch, dist_mod=dist)
st.run(6)
st.sync()
ref = reference_run(sc, 6)
lo, hi = be.begin, be.begin + be.count
ok = (np.array_equal(be.pos[lo:hi], ref['pos'][lo:hi]) and np.array_equal(be.vel[lo:hi], ref['vel'][lo:hi])
      and np.array_equal(be.flags[lo:hi], ref['flags'][lo:hi]) and np.array_equal(be.pos, ref['pos']))
print('RANK', rank, 'OK' if ok else 'MISMATCH', flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


@pytest.mark.parametrize('world', [2, 8])
def test_sharded_stepping_over_gloo_matches_single_process(tmp_path, world):
    """world_size 2 and 8 over gloo: each rank steps its shard with the checker backend, exchanges the moved 48-byte records
    with all_gather_into_tensor exactly as the GPU path does, and must land on the single-process result bit for bit."""
    import subprocess
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    port = str(29531 + world)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=port, OMP_NUM_THREADS='1' if world > 2 else '2')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
                        '--master-addr', '127.0.0.1', '--master-port', port, str(script), ROOT],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count('OK') == world, r.stdout[-2000:]


def test_binvox_map_to_obstacles_matches_reference():
    """exp3 map (the reference's visualization/map/map.binvox, kept as a data fixture) -> the 1491 spheres the reference's
    read_obstacle produced for F10, same order (= obstacle ids), same coordinates bit for bit (mamp/read_map.py:42-85)."""
    import os
    from golden_util import GOLDEN, load
    from sca_amd import read_map, scenarios
    fx = load('F10_sca_exp3_map')
    objs = read_map.read_obstacle(center=(35, 30), environ='exp3', obs_path=os.path.join(GOLDEN, 'exp3_map.binvox'))
    assert len(objs) == 1491 and [o.id for o in objs] == list(range(1491))
    assert np.array_equal(np.array([o.pos_global_frame for o in objs]), fx['obs_pos'])
    assert all(o.radius == 0.2 and o.shape == 'sphere' for o in objs)
    assert read_map.read_obstacle(center=(0, 0), environ='exp1', obs_path='unused') == []
    with open(os.path.join(GOLDEN, 'exp3_map.binvox'), 'rb') as f:
        m = read_map.read_as_3d_array(f, fix_coords=False)
    assert m.axis_order == 'xzy' and list(m.data.shape) == m.dims
    sc = scenarios.spawn_n_drones(16)
    assert np.array_equal(sc['start'], fx['start']) and np.array_equal(sc['goal'], fx['goal6'])


def test_binvox_rejects_garbage(tmp_path):
    from sca_amd import read_map
    p = tmp_path / 'x.binvox'
    p.write_bytes(b'#notbinvox 1\n')
    with pytest.raises(IOError):
        read_map.read_as_3d_array(open(p, 'rb'))
    p.write_bytes(b'#binvox 1\ndim 2 2 2\ntranslate 0 0 0\nscale 1\ndata\n\x01\x07')
    with pytest.raises(IOError):
        read_map.read_as_3d_array(open(p, 'rb'))            # 7 voxels for a 2x2x2 grid
    p.write_bytes(b'#binvox 1\ndim 2 2 2\ntranslate 0 0 0\nscale 1\ndata\n\x01\x03\x00\x05')
    m = read_map.read_as_3d_array(open(p, 'rb'))
    assert m.data.sum() == 3


def test_usable_cores_respects_quota(monkeypatch, tmp_path):
    from sca_amd import hostinfo
    n = hostinfo.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_scenario_generators_match_reference_fixtures():
    """sphere / circle / take-off generators against the start and goal states the reference's own generator functions
    produced for the golden fixtures (run_orca.py:17-54, run_sca.py:17-30,53-81)."""
    from golden_util import load
    from sca_amd import scenarios
    fx = load('F3_srvo_sphere100')
    sc = scenarios.sphere(100)
    assert np.array_equal(sc['start'], fx['start']) and np.array_equal(sc['goal'], fx['goal6'])
    fx = load('F1_sca_circle8')
    sc = scenarios.circle(8, rad=10.0)
    assert np.array_equal(sc['start'], fx['start']) and np.array_equal(sc['goal'], fx['goal6'])
    fx = load('F4_sca_takeoff16')
    sc = scenarios.takeoff_landing(16)
    assert np.array_equal(sc['start'], fx['start']) and np.array_equal(sc['goal'], fx['goal6'])
    assert np.array_equal(sc['obs_pos'], fx['obs_pos'])



def test_bench_refuses_a_gpu_count_that_contradicts_its_launcher():
    """launched BY a launcher (WORLD_SIZE set) with another --gpus: refuse; without a launcher bench.py starts the ranks itself
    (tests/test_gpu_bench.py::test_bench_self_launches_two_ranks)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '1'],
                       env=dict(os.environ, WORLD_SIZE='1'), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and 'torch.distributed.run' in (r.stdout + r.stderr)


def test_bench_self_launch_starts_torch_distributed_run_as_a_child_and_returns_its_code():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (the driver's command shape): the ranks are started as a child job.  No GPU
    here, so both ranks stop at "bench.py needs a GPU" -- what is checked is the plumbing: the launch line, that each rank got
    WORLD_SIZE = 2 (it passed the --gpus check) and that the parent returns the job's non-zero code without touching a GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1', '--agents', '64'],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    both = r.stdout + r.stderr
    assert 'launching' in both and '--nproc-per-node 2' in both
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and ('needs a GPU' in both or 'No HIP GPUs' in both or 'ProcessGroupNCCL' in both or 'NCCL' in both or 'CUDA' in both), both[-3000:]


def test_bench_compact_line_stays_under_4k_and_keeps_the_contract_keys():
    """the driver could not parse round 5's 24-KB stdout line.  bench.py's last stdout line is compact_record(out): run on the
    full record round 5 printed (profiles/r05_a_bench_c4_driver_cmd.json, 23.8 KB) it must stay under 4 KB and carry the contract's
    keys, `roofline` and `cpu_baseline` as flat objects, one number per extra leg -- also when every string in the record is huge."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, 'profiles', 'r05_a_bench_c4_driver_cmd.json')) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000
    line = bench.compact_record(full)
    assert len(line) < 4096 and '\n' not in line
    c = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline', 'rccl_ranks_seen', 'max_abs_dv', 'c2_ms', 'c3_auto_ms', 'c5_ms'):
        assert k in c, k
    assert c['metric'] == full['metric'] and c['steps'] == 20 and c['warmup'] == 5 and c['n_gpus'] == 1 and c['dtype'] == 'f64'
    assert abs(c['value'] - full['value']) < 1e-6 * full['value'] and abs(c['ms_per_step'] - full['ms_per_step']) < 1e-6
    assert {'workload', 'agents', 'agents_per_gpu', 'neighbor_search', 'v_pref', 'parallelism'} <= set(c['config'])
    r = c['roofline']
    assert {'bound', 'kernel', 'kernel_ms', 'achieved', 'peak', 'unit', 'frac', 'bytes_per_unit', 'units_per_launch', 'traffic', 'valu_issue_frac'} <= set(r)
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-6 and r['bound'] == 'hbm'
    b = c['cpu_baseline']
    assert b['kind'] == 'port' and b['cores'] == 16 and b['one_thread']['value'] > 0 and b['reference_python'] == {'value': 35.9, 'cores': 1}
    # nested objects only where asked for; everything else scalar (the driver's parser keeps scalars of sub-objects)
    assert all(not isinstance(v, (dict, list)) for k in ('config', 'roofline') for v in c[k].values())
    # a hostile record: every string 5 KB long
    def blow(o):
        if isinstance(o, dict):
            return {k: blow(v) for k, v in o.items()}
        if isinstance(o, str) and o not in ('hbm', 'port', 'GB/s'):
            return o + ' x' * 2500
        return o
    fat = blow(full)
    fat['metric'], fat['unit'], fat['scaling'], fat['dtype'], fat['data'] = full['metric'], full['unit'], full['scaling'], full['dtype'], full['data']
    assert len(bench.compact_record(fat)) < 4096
