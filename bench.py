#!/usr/bin/env python3
"""bench.py -- agent-steps/sec of the batched velocity-solver hot path on MI355X.

A "step" = one pass of the hot path over the whole swarm: kd-tree neighbour selection -> RVO-cone / ORCA
half-space construction -> 513-candidate sweep + selection (or LP) -> env update (integrate + collision / goal
flags), with the state resident in HBM when the timed region starts.

  python bench.py --gpus 1 --steps K --warmup W            (N>1: launched by torch.distributed.run, one rank per GPU)

Workloads (BASELINE.json configs): c4 = circle N=100000 SCA (default: the configuration the metric's "N-agent circle at
1/2/4/8 GPUs" clause names; it fits one GPU), c2 = circle N=1024 SCA, c3 = random N=4096 ORCA3D, c5 = take-off/landing
N=16384 mixed SCA + S-RVO3D.  With several GPUs the default is WEAK scaling: the workload's agent count per GPU (one
circle of 100000 x n_gpus agents, sharded by id, one all-gather of the moved 48-byte records per step);
`--scaling strong` keeps the total at the workload's N.
SCA's preferred velocity comes from the reference's Dubins tracker (scaPolicy.py:264-338), which is outside the path
north_star names: by default the bench feeds the straight-line rule (rvo3dPolicy.py:182-196) computed on
the device instead, and says so in `config`; at --gpus 1 (or with --end-to-end) a second timed leg, `end_to_end_sca`, repeats
the steps with the tracker on the device.  `--vpref dubins-device` runs the tracker on the device inside every step
(end-to-end SCA, state still resident); `--vpref dubins` runs the native host tracker (bit-exact, host-bound).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
# algorithmic bytes per agent-step, fp64-position records: 48 B own record + 32 B private inputs
# + 16 neighbours x 48 B + 32 B action row
BYTES_PER_AGENT_STEP = 48 + 32 + 16 * 48 + 32

WORKLOADS = {
    'c2': dict(kind='circle', n=1024, policy='sca', desc='c2: circle N=1024, SCA policy'),
    'c3': dict(kind='random', n=4096, policy='orca', desc='c3: random N=4096, ORCA3D policy (sampled, as run_orca.py)'),
    'c3lp': dict(kind='random', n=4096, policy='orcalp', desc='c3: random N=4096, ORCA3D Official (LP1-4)'),
    'c4': dict(kind='circle', n=100000, policy='sca', desc='c4: circle N=100000, SCA policy'),
    'c5': dict(kind='takeoff', n=16384, policy='mixed', desc='c5: take-off/landing N=16384, SCA even ids / S-RVO3D odd ids'),
}
POL = {'sca': 0, 'rvo': 1, 'srvo': 2, 'orca': 3, 'orcalp': 4}


def build_scene(w, n):
    from sca_amd import scenarios, solver as S
    if w['kind'] == 'circle':
        sc = scenarios.circle(n)
    elif w['kind'] == 'random':
        sc = scenarios.random_cube(n, seed=0)
    else:
        sc = scenarios.takeoff_landing(n)
    n = len(sc['start'])
    if w['policy'] == 'mixed':
        policy = np.where(np.arange(n) % 2 == 0, 0, 2).astype(np.uint8)
    else:
        policy = np.full(n, POL[w['policy']], np.uint8)
    return dict(n=n, sc=sc, policy=policy, radius=np.full(n, 0.5), pref_speed=np.full(n, 1.0),
                zaxis=S.zaxis_flags(sc['start'], sc['goal']), max_run_dist=scenarios.max_run_dist(sc['start'], sc['goal']))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default=None, choices=sorted(WORKLOADS))
    ap.add_argument('--agents', type=int, default=None)
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='N>1 GPUs: weak = the workload\'s agent count PER GPU (default), strong = the same total')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-end-to-end', action='store_true',
                    help='skip the second timed leg (SCA workloads: the same steps with the Dubins v_pref tracker on the device)')
    ap.add_argument('--end-to-end', action='store_true',
                    help='run the second leg with several GPUs too (by default only at --gpus 1: the scaling runs measure `value`)')
    ap.add_argument('--vpref', default='straight', choices=['straight', 'dubins', 'dubins-device'],
                    help='dubins: SCA v_pref from the native host-side tracker every step (end-to-end SCA, host-bound); '
                         'dubins-device: from the device tracker inside every step (end-to-end SCA, resident)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    import torch
    dist = None
    # test hook: all ranks on GPU 0 with a host-staged gloo exchange, to exercise this script's N > 1 path on a 1-GPU box
    share_gpu = bool(os.environ.get('SCA_BENCH_SHARE_GPU'))
    if share_gpu:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        if share_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: libsca_hip has no CPU path')

    from sca_amd import solver as S
    from sca_amd.distributed import ShardedStepper

    wname = args.workload or 'c4'
    w = WORKLOADS[wname]
    n_req = args.agents or w['n']
    if world > 1:
        if args.scaling == 'weak':
            n_req *= world                                     # per-GPU work fixed: N grows with the GPU count
        n_req = ((n_req + world - 1) // world) * world
    scene = build_scene(w, n_req)
    n = scene['n']
    sc = scene['sc']

    sol = S.BatchedSolver(max_agents=n, max_obstacles=max(1, len(sc['obs_radius'])), device=local_rank)
    sol.set_obstacles(sc['obs_pos'], sc['obs_radius'])
    sol.set_agents(scene['radius'], scene['pref_speed'], sc['goal'][:, :3], scene['policy'], scene['zaxis'],
                   scene['max_run_dist'])
    sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
    stepper = ShardedStepper(sol, rank, world, torch_mod=torch, dist_mod=dist, staged=share_gpu)

    if args.vpref == 'dubins-device':
        sol.device_tracker_enable(sc['goal'][:, 3:6])
    if args.vpref == 'dubins':
        if world > 1:
            raise SystemExit('--vpref dubins is a single-GPU measurement')
        from sca_amd import tracker as trk
        tr = trk.DubinsTracker(sc['goal'][:, :3], sc['goal'][:, 3:6], scene['pref_speed'], scene['zaxis'])
        ext = np.isin(scene['policy'], (0, 5))

        class TrackedStepper:
            """one step = read back state -> native tracker (host threads) -> upload v_pref -> resident GPU step"""

            def run(self, k):
                for _ in range(k):
                    st = sol.get_state()
                    active = ((st['flags'] & 7) == 0) & ext
                    vp = tr.vpref(st['pos'], st['vel'], st['heading'], active.astype(np.uint8))
                    sol.set_vpref(vp, ext.astype(np.uint8))
                    sol.run_steps(1)
                    tr.note_nbr0(sol.nbr0())

            def sync(self):
                sol.synchronize()
        stepper = TrackedStepper()
    # warm-up (untimed): includes the bootstrap step (velocity 0 -> 0.3 v_pref) so the real branch runs afterwards
    stepper.run(args.warmup)
    stepper.sync()
    sol.agent_steps(reset=True)
    sol.set_profiling(True)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    stepper.run(args.steps)
    stepper.sync()
    barrier()
    dt = time.perf_counter() - t0
    sol.set_profiling(False)
    my_steps = sol.agent_steps(reset=True)
    kms = sol.kernel_ms()
    if dist is not None:
        red_dev = 'cpu' if share_gpu else 'cuda'
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([my_steps], dtype=torch.int64, device=red_dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total_steps = int(c.item())
    else:
        total_steps = my_steps

    cpu = None
    if rank == 0 and not args.no_cpu_baseline and world == 1:   # the CPU leg is a single-GPU (rank 0, N = 1) measurement
        cpu = cpu_baseline(scene, sol, S, tracked=args.vpref != 'straight', warmup=max(2, args.warmup))

    # second leg, SCA workloads only: the same step with SCA's own v_pref -- the Dubins tracker of scaPolicy.py:264-338 as
    # kernels inside every step (SURVEY.md 8d asks for both the solver and the end-to-end throughput)
    e2e = None
    if args.vpref == 'straight' and not args.no_end_to_end and w['policy'] in ('sca', 'mixed') and (world == 1 or args.end_to_end):
        # from the start state again, so that both legs time the same stretch of the episode
        sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
        sol.device_tracker_enable(sc['goal'][:, 3:6])
        stepper.run(args.warmup)
        stepper.sync()
        sol.agent_steps(reset=True)
        barrier()
        t0 = time.perf_counter()
        stepper.run(args.steps)
        stepper.sync()
        barrier()
        dt2 = time.perf_counter() - t0
        steps2 = sol.agent_steps(reset=True)
        replans = int(sol.device_tracker_replans()[stepper.begin:stepper.begin + stepper.count].sum())
        if dist is not None:
            t = torch.tensor([dt2], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = float(t.item())
            c = torch.tensor([steps2, replans], dtype=torch.int64, device=red_dev)
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            steps2, replans = int(c[0].item()), int(c[1].item())
        e2e = {'value': steps2 / dt2, 'unit': 'agent-steps/s', 'ms_per_step': dt2 / args.steps * 1e3,
               'v_pref': 'Dubins tracker + 3-D Dubins planner on the device inside every step (k_track, k_replan / k_replan_few)',
               'plans_since_enable': replans, 'agent_steps_timed': steps2}

    if rank == 0:
        value = total_steps / dt
        per_launch_agents = my_steps / max(args.steps, 1)
        solve_s = kms['solve'] * 1e-3
        achieved = (BYTES_PER_AGENT_STEP * per_launch_agents / solve_s / 1e9) if solve_s > 0 else 0.0
        out = {
            'metric': 'agent_steps_per_sec', 'value': value, 'unit': 'agent-steps/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': w['desc'] + (f' [--agents {args.agents}]' if args.agents else '')
                       + (f' -- weak scaling: {w["n"] if not args.agents else args.agents} agents per GPU'
                          if world > 1 and args.scaling == 'weak' else ''),
                       'agents': n, 'agents_per_gpu': n // world, 'neighbor_search': 'kd-tree of kdTree.py rebuilt on the device every step (replicated per rank), device query',
                       'v_pref': {'straight': 'straight-line rule on device (SCA\'s Dubins tracker is outside the path: --vpref)',
                                  'dubins': 'native Dubins tracker on the host every step (end-to-end SCA)',
                                  'dubins-device': 'Dubins tracker on the device inside every step (end-to-end SCA)'}[args.vpref],
                       'parallelism': f'agents sharded over {world} GPU(s), all-gather of 48-B records per step'
                       if world > 1 else 'single GPU', 'agent_steps_timed': total_steps},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': measured_traffic(wname), 'kernel': 'k_solve',
                         'kernel_ms': kms['solve'],
                         # with the tracker's re-plans on a side stream the interval before k_solve also holds the join
                         'neighbors_kernel_ms': kms['neighbors'] if args.vpref != 'dubins-device' else None,
                         'bytes_per_agent_step': BYTES_PER_AGENT_STEP,
                         'valu_issue_frac': valu_issue_frac(wname, per_launch_agents, solve_s),
                         'note': 'fp64 VALU bound (no contraction, no MFMA): HBM fraction is reported as required'},
        }
        if e2e is not None:
            out['end_to_end_sca'] = e2e
        if cpu is not None:
            out['cpu_baseline'] = cpu
        print(json.dumps(out), flush=True)
    sol.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


VALU_PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 4      # SIMDs x clock / cycles per wave64 fp64 VALU instruction


def valu_issue_frac(wname, agents_per_launch, kernel_s):
    """k_solve is bound by fp64 VALU issue, not by HBM: fraction of the chip's VALU issue rate it sustains, from the PMC
    instruction count per agent (profiles/r01_pmc_traffic.json) and the launch duration measured in this run."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
    try:
        with open(path) as f:
            per_agent = json.load(f).get(wname, {}).get('k_solve_valu_wave_insts_per_agent')
    except (OSError, ValueError):
        per_agent = None
    if not per_agent or kernel_s <= 0:
        return None
    return per_agent * agents_per_launch / kernel_s / VALU_PEAK_WAVE_INSTS


def measured_traffic(wname):
    """HBM bytes per k_solve launch from the rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate runs,
    profiles/r01_pmc_traffic.json); None when no capture exists for this workload."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
    try:
        with open(path) as f:
            return json.load(f).get(wname, {}).get('k_solve_hbm_bytes_per_launch')
    except (OSError, ValueError):
        return None


def cpu_baseline(scene, sol, S, tracked=False, warmup=20):
    """The CPU oracle (decision-identical C restatement of the reference, oracle/sca_oracle.c) timed on this box's host
    cores on a bounded sample: policy passes over the current device state.  Also reports max |v_hip - v_oracle|."""
    from oracle import oracle as orc
    n = scene['n']
    sc = scene['sc']
    st = sol.get_state()
    sample_state = 'after the timed steps'
    if not ((st['flags'] & 7) == 0).any():
        # short episodes (take-off / landing: 10 m apart) are over by now: sample the state after the warm-up steps instead
        sol.set_state(sc['start'][:, :3], np.zeros((n, 3), np.float32), sc['start'][:, 3:6], np.zeros(n, np.uint8))
        sol.run_steps(warmup)
        sol.synchronize()
        st = sol.get_state()
        sample_state = f'after {warmup} steps from the start (the episode is over after the timed steps)'
    from sca_amd import hostinfo
    cores = min(hostinfo.usable_cores(), 64)            # affinity AND cgroup quota: threads beyond it only oversubscribe
    perm = sol.get_kd_perm()
    # the same pass on the GPU for the parity number (first: with a tracker the oracle is fed the v_pref this pass used)
    sol.policy_pass(S.NBR_KDTREE)
    a = sol.actions()
    vused = np.nan_to_num(sol.diag()['vpref'])
    vmode = (np.isin(scene['policy'], (0, 5)).astype(np.uint8) if tracked else np.zeros(n, np.uint8))
    args = (st['pos'], st['vel'], st['heading'], scene['radius'], scene['pref_speed'], st['flags'], sc['goal'][:, :3],
            scene['policy'], scene['zaxis'], vused, vmode, perm, sc['obs_pos'], sc['obs_radius'])
    t0 = time.perf_counter()
    ref = orc.policy_step(*args, nthreads=cores)
    reps = 1
    one = time.perf_counter() - t0
    while time.perf_counter() - t0 < 10.0 and reps < 50:
        orc.policy_step(*args, nthreads=cores)
        reps += 1
    dt = time.perf_counter() - t0
    active = int(((st['flags'] & 7) == 0).sum())
    dv = float(np.abs(a[:, :3] - ref['action'][:, :3]).max())
    return {'value': active * reps / dt, 'unit': 'agent-steps/s', 'cores': cores, 'kind': 'port',
            'sample': f'{reps} policy passes over the {n}-agent state {sample_state} ({one:.2f} s each), '
                      f'OpenMP over agents', 'max_abs_dv_vs_hip': dv}


if __name__ == '__main__':
    main()
