"""bench.py's own code paths on the GPU box: the JSON contract of a single-GPU run (value_parity, cpu_baseline incl. the tracker
pass), the --vpref dubins leg (it died with a KeyError behind the timed run), and the script as the driver launches it
for eight ranks, sharing GPU 0 through its test hook so that the first real SCALE run cannot die on plumbing."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


COMPACT_KEYS = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
                'data', 'config', 'roofline', 'rccl_ranks_seen', 'max_abs_dv', 'detail'}


def _run(args, env=None, timeout=900, tmp=None):
    """-> the full record.  stdout holds ONE line, the compact record (< 4 KB: the driver's parser lost round 5's 24-KB line); the full
    record is the file SCA_BENCH_DETAIL names and the stderr line prefixed DETAIL -- both are read and must agree with the compact one."""
    import tempfile
    detail = os.path.join(tmp or tempfile.mkdtemp(prefix='sca_bench_'), 'bench_detail.json')
    r = subprocess.run([sys.executable] + args, env=dict(os.environ, SCA_BENCH_DETAIL=detail, **(env or {})), capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    nonempty = [l for l in r.stdout.splitlines() if l.strip()]
    lines = [l for l in nonempty if l.startswith('{')]                # (gloo prints its own "[Gloo] Rank ..." lines to stdout)
    assert len(lines) == 1 and nonempty[-1] == lines[0], r.stdout[-2000:]     # ONE JSON line, and it is the LAST line
    assert len(lines[0]) < 4096, len(lines[0])
    compact = json.loads(lines[0])
    assert COMPACT_KEYS <= set(compact), sorted(COMPACT_KEYS - set(compact))
    with open(detail) as f:
        full = json.load(f)
    err = [l for l in r.stderr.splitlines() if l.startswith('DETAIL ')]
    assert len(err) == 1 and json.loads(err[0][7:]) == full
    for k in ('metric', 'unit', 'n_gpus', 'steps', 'warmup', 'scaling', 'dtype', 'data', 'higher_is_better'):
        assert compact[k] == full[k], k
    assert abs(compact['value'] - full['value']) <= 1e-6 * full['value'] and abs(compact['ms_per_step'] - full['ms_per_step']) <= 1e-5 * full['ms_per_step']
    full['_compact'] = compact
    return full


def test_single_gpu_json_line_small_swarm():
    out = _run([os.path.join(ROOT, 'bench.py'), '--agents', '6000', '--steps', '5', '--warmup', '3'])
    assert out['metric'] == 'agent_steps_per_sec' and out['n_gpus'] == 1 and out['dtype'] == 'f64' and out['vs_baseline'] is None
    assert out['config']['agents'] == 6000 and out['config']['re_plans_timed'] > 0
    r = out['roofline']
    assert r['bound'] == 'hbm' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    vp = out['value_parity']                                    # SCA as shipped against the host-tracker run: identical
    assert vp['deviating_agent_steps'] == 0 and vp['max_abs_dv'] == 0.0 and vp['re_plan_counts_equal'] and vp['agent_steps'] > 0
    cpu = out['cpu_baseline']
    assert cpu['kind'] == 'port' and cpu['max_abs_dv_vs_hip_solver_given_vpref'] == 0.0
    assert cpu['with_tracker']['value'] < cpu['policy_only']['value'] and cpu['value'] == cpu['with_tracker']['value']
    assert 'solver_only' in out and 'grid_mode' in out and 'scale_model' in out
    # the compact line: the record the driver parses carries the roofline and the CPU baseline as flat objects
    c = out['_compact']
    assert c['config']['agents'] == 6000 and c['config']['workload'].startswith('c4')
    cr = c['roofline']
    assert {'bound', 'kernel', 'kernel_ms', 'achieved', 'peak', 'unit', 'frac', 'bytes_per_unit', 'units_per_launch', 'traffic'} <= set(cr)
    assert cr['bound'] == 'hbm' and abs(cr['frac'] - cr['achieved'] / cr['peak']) < 1e-5 * cr['frac'] + 1e-12
    cc = c['cpu_baseline']
    assert cc['kind'] == 'port' and cc['cores'] >= 1 and cc['value'] > 0 and cc['one_thread']['value'] > 0 and cc['reference_python']['value'] > 0
    assert c['max_abs_dv'] == 0.0 and c['rccl_ranks_seen'] == 1
    assert all(not isinstance(v, (dict, list)) for k in ('config', 'roofline') for v in c[k].values())


def test_host_tracker_leg_prints_its_line():
    out = _run([os.path.join(ROOT, 'bench.py'), '--workload', 'c2', '--vpref', 'dubins', '--steps', '4', '--warmup', '3', '--no-cpu-baseline'])
    assert out['value'] > 0 and 'host' in out['config']['v_pref'] and out['config']['kernel_forms']


def test_bench_script_eight_ranks_on_one_gpu():
    """torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 with all ranks on GPU 0 (SCA_BENCH_SHARE_GPU: gloo group,
    host-staged exchange): the launch line, the rendezvous, the strong-scaling shard arithmetic, the reductions and the one JSON
    line are the ones the 8-GPU run uses."""
    out = _run(['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1', '--master-port',
                '29551', os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '4', '--warmup', '3', '--agents', '8000'],
               env={'SCA_BENCH_SHARE_GPU': '1', 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29551'}, timeout=1500)
    assert out['n_gpus'] == 8 and out['scaling'] == 'strong' and out['config']['agents'] == 8000 and out['config']['agents_per_gpu'] == 1000
    assert out['config']['agent_steps_timed'] == 8000 * 4 and out['value'] > 0


def test_bench_script_partition_mode_two_ranks_on_one_gpu():
    """bench.py --gpus 2 --nbr grid --partition (cell-owner partition, point-to-point halo exchange) through the same test hook"""
    out = _run(['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
                '29553', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '3', '--agents', '6000', '--nbr', 'grid',
                '--partition'], env={'SCA_BENCH_SHARE_GPU': '1', 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29553'}, timeout=1500)
    assert out['n_gpus'] == 2 and out['config']['agents'] == 6000 and 'slabs of grid cells' in out['config']['parallelism']
    assert out['config']['agent_steps_timed'] == 6000 * 5 and out['value'] > 0


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset -- the command shape the driver's 1-GPU record shows -- must start its own
    torch.distributed.run job as a child, relay rank 0's line and say how many ranks the process group reported."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env['SCA_BENCH_SHARE_GPU'] = '1'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '3', '--agents', '6000'],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and len(lines[0]) < 4096, r.stdout[-2000:]
    assert json.loads(lines[0])['rccl_ranks_seen'] == 2 and json.loads(lines[0])['exchange'] == 'torch'
    out = json.loads([l for l in r.stderr.splitlines() if l.startswith('DETAIL ')][0][7:])
    assert out['n_gpus'] == 2 and out['rccl_ranks_seen'] == 2 and out['process_group']['backend'] == 'gloo'
    assert out['config']['agent_steps_timed'] == 6000 * 4 and out['value'] > 0


def test_bench_self_launch_partition_variant():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env['SCA_BENCH_SHARE_GPU'] = '1'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '3', '--agents', '6000',
                        '--nbr', 'grid', '--partition'], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    c = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert c['rccl_ranks_seen'] == 2 and c['exchange'] == 'partition'
    out = json.loads([l for l in r.stderr.splitlines() if l.startswith('DETAIL ')][0][7:])
    assert out['rccl_ranks_seen'] == 2 and out['process_group']['exchange'] == 'partition'


def test_env_api_key_times_the_drop_in_loop():
    """`env_api` (round 5): the drop-in MACAEnv built from Agent objects, `while not env.step()` -- three variants per config (step alone,
    + the action rows read back, + every Agent's pos_global_frame read), the resident burst of the same scene beside them."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--env-api-only', 'c2,c3', '--steps', '20', '--warmup', '5'],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])['env_api']
    for name, n in (('c2', 1024), ('c3', 4096)):
        row = out[name]
        assert row['agents'] == n and row['device_tracker'] == (name == 'c2')
        for v in ('step', 'step_actions', 'step_agent_pos'):
            assert row[v]['ms_per_step'] > 0 and row[v]['value'] > 0 and row[v]['steps'] > 0
        assert row['step']['ms_per_step'] <= row['step_agent_pos']['ms_per_step']
        assert 0.8 < row['step_over_resident'] < 4.0, row           # the loop costs the kernels + one synchronising read-back per step
        # the boundary with host buffers on both sides of every step (INTEGRATION.md stub B): the PCIe-inclusive rate, beside the resident one
        hh = row['host_handover']
        assert hh['ms_per_step'] > row['resident_ms_per_step'] and hh['value'] > 0 and hh['steps'] == 20
        assert hh['host_to_device_bytes_per_step'] == n * 73 and hh['device_to_host_bytes_per_step'] == n * 101


def test_first_multi_gpu_script_plumbing(tmp_path):
    """tools/gpu/first_multi_gpu.sh (what the first lease with >= 2 GPUs runs: the RCCL tests, bench.py --gpus G in both exchange forms, the
    partition variant, one JSON) with all ranks on GPU 0 through the bench's test hook: the script itself is known to work."""
    out = tmp_path / 'first'
    env = dict(os.environ, SCA_BENCH_SHARE_GPU='1', SCA_FIRST_GPUS='2', SCA_FIRST_STEPS='4', SCA_FIRST_WARMUP='3', SCA_FIRST_AGENTS='6000',
               GRAFT_REPO_ROOT=ROOT)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'gpu', 'first_multi_gpu.sh'), str(out)], env=env, capture_output=True, text=True,
                       timeout=2400, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rec = json.load(open(out / 'first_multi_gpu.json'))
    rows = {row['file']: row for row in rec['runs']}
    assert set(rows) == {'g2_allgather_detail.json', 'g2_partition_detail.json'}, r.stdout[-3000:]
    for row in rows.values():
        assert row['n_gpus'] == 2 and row['rccl_ranks_seen'] == 2 and row['value'] > 0 and row['agents'] == 6000
        assert row['ranks_sharing_gpu0_test_hook'] is True and row['backend'] == 'gloo'
    assert rows['g2_partition_detail.json']['exchange'] == 'partition' and rows['g2_allgather_detail.json']['exchange'] == 'torch'
    assert 'rccl tests rc' in (out / 'summary.txt').read_text()
