"""BatchedSolver: numpy-facing wrapper of one libsca_hip context (include/sca_hip.h).

All heavy lifting happens in the HIP kernels; this class only marshals arrays.  It mirrors the quantities the
reference keeps on its Agent objects (mamp/agents/agent.py) as structure-of-arrays.
"""
import ctypes as C
import math

import numpy as np

from . import _lib

POL_SCA, POL_RVO3D, POL_SRVO3D, POL_ORCA3D, POL_ORCA3D_LP, POL_RVO3D_DUBINS = range(6)
FLAG_AT_GOAL, FLAG_COLLISION, FLAG_TIMEOUT = 1, 2, 4
NBR_KDTREE, NBR_GRID, NBR_KDTREE_HOSTBUILD, NBR_AUTO = 0, 1, 2, 3
FORM_SOLVE_SPLIT, FORM_TRACK_FUSED, FORM_REPLAN_LANE, FORM_REPLAN_FEW, FORM_LP_LANE, FORM_SOLVE_FB, FORM_ACTION_FB, FORM_AUTO_TAIL = 1, 2, 4, 8, 16, 32, 64, 128   # sca_last_pass_forms
K = _lib.K


class ScaError(RuntimeError):
    pass


class BatchedSolver:
    def __init__(self, max_agents, max_obstacles=0, device=0, params=None):
        self.L = _lib.lib()
        p = _lib.Params()
        self.L.sca_default_params_v2(C.byref(p), C.sizeof(p))
        for k, v in (params or {}).items():
            setattr(p, k, v)
        self.params = p
        self.ctx = C.c_void_p()
        rc = self.L.sca_create(C.byref(p), int(device), int(max_agents), int(max_obstacles), C.byref(self.ctx))
        if rc != 0:
            msg = self.L.sca_last_error(self.ctx).decode() if self.ctx else 'sca_create failed'
            if self.ctx:
                self.L.sca_destroy(self.ctx)
                self.ctx = None
            raise ScaError(f'sca_create: {msg} (rc={rc})')
        self.n = 0
        self.m = 0

    def close(self):
        if getattr(self, 'ctx', None):
            self.L.sca_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise ScaError(f'{what}: {self.L.sca_last_error(self.ctx).decode()} (rc={rc})')

    # ---- static scene ------------------------------------------------------------------------------
    def set_obstacles(self, pos, radius):
        pos = _lib.as_d(pos).reshape(-1, 3)
        radius = _lib.as_d(radius).reshape(-1)
        self.m = len(radius)
        self._chk(self.L.sca_set_obstacles(self.ctx, self.m, _lib.ptr(pos, C.c_double), _lib.ptr(radius, C.c_double)),
                  'sca_set_obstacles')

    def set_agents(self, radius, pref_speed, goal, policy, zaxis=None, max_run_dist=None):
        radius = _lib.as_d(radius).reshape(-1)
        n = len(radius)
        pref_speed = _lib.as_d(np.broadcast_to(pref_speed, (n,)))
        goal = _lib.as_d(goal).reshape(n, 3)
        policy = np.ascontiguousarray(np.broadcast_to(policy, (n,)), np.uint8)
        zaxis = np.zeros(n, np.uint8) if zaxis is None else np.ascontiguousarray(zaxis, np.uint8)
        mrd = np.full(n, np.inf) if max_run_dist is None else _lib.as_d(max_run_dist).reshape(n)
        self.n = n
        self._chk(self.L.sca_set_agents(self.ctx, n, _lib.ptr(radius, C.c_double), _lib.ptr(pref_speed, C.c_double),
                                        _lib.ptr(goal, C.c_double), _lib.ptr(policy, C.c_uint8),
                                        _lib.ptr(zaxis, C.c_uint8), _lib.ptr(mrd, C.c_double)), 'sca_set_agents')

    def set_agent_params(self, neighbor_dist=None, max_neighbors=None, time_step=None, time_horizon=None, max_speed=None,
                         max_heading_change=None, dt_nominal=None):
        """The solver attributes per agent (the reference keeps them on every Agent object, agent.py:24-41): arrays of n, None = the context's
        value for everybody; with no argument: back to one value per context.  After set_agents, before device_tracker_enable."""
        n = self.n
        keep = []

        def arr(a, dt, ct):
            if a is None:
                return None
            b = np.ascontiguousarray(np.broadcast_to(a, (n,)), dt)
            keep.append(b)
            return _lib.ptr(b, ct)
        args = [arr(neighbor_dist, np.float64, C.c_double), arr(max_neighbors, np.int32, C.c_int32), arr(time_step, np.float64, C.c_double),
                arr(time_horizon, np.float64, C.c_double), arr(max_speed, np.float64, C.c_double), arr(max_heading_change, np.float64, C.c_double),
                arr(dt_nominal, np.float64, C.c_double)]
        self._chk(self.L.sca_set_agent_params(self.ctx, n if keep else 0, *args), 'sca_set_agent_params')

    # ---- dynamic state -------------------------------------------------------------------------------
    def set_state(self, pos, vel, heading, flags, total_dist=None, step_num=None):
        n = self.n
        pos = _lib.as_d(pos).reshape(n, 3)
        vel = np.ascontiguousarray(vel, np.float32).reshape(n, 3)
        heading = _lib.as_d(heading).reshape(n, 3)
        flags = np.ascontiguousarray(flags, np.uint8).reshape(n)
        td = None if total_dist is None else _lib.as_d(total_dist).reshape(n)
        sn = None if step_num is None else np.ascontiguousarray(step_num, np.int32).reshape(n)
        self._chk(self.L.sca_set_state(self.ctx, _lib.ptr(pos, C.c_double), _lib.ptr(vel, C.c_float),
                                       _lib.ptr(heading, C.c_double), _lib.ptr(flags, C.c_uint8),
                                       None if td is None else _lib.ptr(td, C.c_double),
                                       None if sn is None else _lib.ptr(sn, C.c_int32)), 'sca_set_state')

    def get_state(self):
        n = self.n
        out = dict(pos=np.zeros((n, 3)), vel=np.zeros((n, 3), np.float32), heading=np.zeros((n, 3)),
                   flags=np.zeros(n, np.uint8), total_dist=np.zeros(n), step_num=np.zeros(n, np.int32))
        self._chk(self.L.sca_get_state(self.ctx, _lib.ptr(out['pos'], C.c_double), _lib.ptr(out['vel'], C.c_float),
                                       _lib.ptr(out['heading'], C.c_double), _lib.ptr(out['flags'], C.c_uint8),
                                       _lib.ptr(out['total_dist'], C.c_double), _lib.ptr(out['step_num'], C.c_int32)),
                  'sca_get_state')
        return out

    def set_kd_perm(self, perm):
        perm = np.ascontiguousarray(perm, np.int32).reshape(self.n)
        self._chk(self.L.sca_set_kd_perm(self.ctx, _lib.ptr(perm, C.c_int32)), 'sca_set_kd_perm')

    def get_kd_perm(self):
        perm = np.zeros(self.n, np.int32)
        self._chk(self.L.sca_get_kd_perm(self.ctx, _lib.ptr(perm, C.c_int32)), 'sca_get_kd_perm')
        return perm

    def get_kd_tree(self):
        t = np.zeros((2 * self.n - 1, 10))
        self._chk(self.L.sca_get_kd_tree(self.ctx, _lib.ptr(t, C.c_double)), 'sca_get_kd_tree')
        return t

    def set_vpref(self, vpref, mode):
        vpref = _lib.as_d(np.nan_to_num(vpref)).reshape(self.n, 3)
        mode = np.ascontiguousarray(np.broadcast_to(mode, (self.n,)), np.uint8)
        self._chk(self.L.sca_set_vpref(self.ctx, _lib.ptr(vpref, C.c_double), _lib.ptr(mode, C.c_uint8)), 'sca_set_vpref')

    # ---- SCA's v_pref tracker on the device (scaPolicy.py:264-338) ---------------------------------------
    def device_tracker_enable(self, goal_heading, turning_radius=1.5, pitchlims=(-math.pi / 4, math.pi / 4), in_pass=True):
        """From now on the SCA / RVO3D+Dubins agents take v_pref from the device tracker: inside every policy pass
        (in_pass=True) or only when device_tracker_vpref() is called."""
        gh = _lib.as_d(goal_heading).reshape(self.n, 3)
        self._chk(self.L.sca_device_tracker_enable(self.ctx, _lib.ptr(gh, C.c_double), float(turning_radius), float(pitchlims[0]),
                                                   float(pitchlims[1]), int(bool(in_pass))), 'sca_device_tracker_enable')

    def device_tracker_set_agent_params(self, turning_radius=None, pitch_lo=None, pitch_hi=None):
        """agent.turning_radius / agent.pitchlims per agent (arrays of n; None = device_tracker_enable's value); after device_tracker_enable"""
        keep = []

        def arr(a):
            if a is None:
                return None
            b = _lib.as_d(np.broadcast_to(a, (self.n,)))
            keep.append(b)
            return _lib.ptr(b, C.c_double)
        self._chk(self.L.sca_device_tracker_set_agent_params(self.ctx, self.n, arr(turning_radius), arr(pitch_lo), arr(pitch_hi)),
                  'sca_device_tracker_set_agent_params')

    def device_tracker_disable(self):
        self._chk(self.L.sca_device_tracker_disable(self.ctx), 'sca_device_tracker_disable')

    def device_tracker_vpref(self, nbr0_dsq=None):
        """One compute_v_pref for every active tracked agent on the current state; nbr0_dsq[i] = distSq of agent.neighbors[0]
        of the previous pass (negative: empty), None = taken from the neighbour lists on the device."""
        out = np.zeros((self.n, 3))
        nb = None if nbr0_dsq is None else _lib.as_d(nbr0_dsq).reshape(self.n)
        self._chk(self.L.sca_device_tracker_vpref(self.ctx, None if nb is None else _lib.ptr(nb, C.c_double),
                                                  _lib.ptr(out, C.c_double)), 'sca_device_tracker_vpref')
        return out

    def device_tracker_replans(self):
        r = np.zeros(self.n, np.int32)
        self._chk(self.L.sca_device_tracker_replans(self.ctx, _lib.ptr(r, C.c_int32)), 'sca_device_tracker_replans')
        return r

    # ---- hot path --------------------------------------------------------------------------------------
    def policy_pass(self, mode=NBR_KDTREE):
        self._chk(self.L.sca_policy_pass(self.ctx, int(mode)), 'sca_policy_pass')

    def env_update(self, want_done=True):
        done = C.c_int(0)
        self._chk(self.L.sca_env_update(self.ctx, C.byref(done) if want_done else None), 'sca_env_update')
        return bool(done.value)

    def run_steps(self, steps, mode=NBR_KDTREE):
        self._chk(self.L.sca_run_steps(self.ctx, int(steps), int(mode)), 'sca_run_steps')

    def env_step(self, mode=NBR_KDTREE):
        """One resident step + the agents still running after it (MACAEnv.step in one library call; synchronises)."""
        v = C.c_int(0)
        self._chk(self.L.sca_env_step(self.ctx, int(mode), C.byref(v)), 'sca_env_step')
        return int(v.value)

    def synchronize(self):
        self._chk(self.L.sca_synchronize(self.ctx), 'sca_synchronize')

    def active_count(self):
        """Agents of this rank not yet done after the last env update (0 == MACAEnv.is_done); synchronises."""
        v = C.c_int(0)
        self._chk(self.L.sca_active_count(self.ctx, C.byref(v)), 'sca_active_count')
        return int(v.value)

    def actions(self):
        a = np.zeros((self.n, 7), np.float32)
        self._chk(self.L.sca_get_actions(self.ctx, _lib.ptr(a, C.c_float)), 'sca_get_actions')
        return a

    def neighbors(self):
        n = self.n
        out = dict(nbr_n=np.zeros(n, np.int32), nbr_id=np.zeros((n, K), np.int32), nbr_kind=np.zeros((n, K), np.uint8),
                   nbr_dsq=np.zeros((n, K)), nbr_valid=np.zeros(n, np.uint8))
        self._chk(self.L.sca_get_neighbors(self.ctx, _lib.ptr(out['nbr_n'], C.c_int32), _lib.ptr(out['nbr_id'], C.c_int32),
                                           _lib.ptr(out['nbr_kind'], C.c_uint8), _lib.ptr(out['nbr_dsq'], C.c_double),
                                           _lib.ptr(out['nbr_valid'], C.c_uint8)), 'sca_get_neighbors')
        return out

    def nbr0(self):
        a = np.zeros(self.n)
        self._chk(self.L.sca_get_nbr0(self.ctx, _lib.ptr(a, C.c_double)), 'sca_get_nbr0')
        return a

    def diag(self):
        n = self.n
        out = dict(diag=np.zeros((n, 5), np.int32), status=np.zeros(n, np.int32), vpref=np.zeros((n, 3)))
        self._chk(self.L.sca_get_diag(self.ctx, _lib.ptr(out['diag'], C.c_int32), _lib.ptr(out['status'], C.c_int32),
                                      _lib.ptr(out['vpref'], C.c_double)), 'sca_get_diag')
        return out

    def kernel_ms(self):
        a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
        self._chk(self.L.sca_last_kernel_ms(self.ctx, C.byref(a), C.byref(b), C.byref(c)), 'sca_last_kernel_ms')
        return dict(neighbors=a.value, solve=b.value, update=c.value)

    def replan_ms(self):
        a = C.c_float(0)
        self._chk(self.L.sca_last_replan_ms(self.ctx, C.byref(a)), 'sca_last_replan_ms')
        return a.value

    def auto_stats(self, reset=False):
        """SCA_NBR_AUTO since the last reset: passes, agents listed for the kd query per pass (mean, max), share of passes with a list"""
        a = (C.c_int64 * 4)()
        self._chk(self.L.sca_auto_stats(self.ctx, a, 1 if reset else 0), 'sca_auto_stats')
        p = max(int(a[0]), 1)
        return {'auto_passes': int(a[0]), 'listed_per_pass_mean': a[1] / p, 'listed_per_pass_max': int(a[2]), 'passes_with_a_list_frac': a[3] / p}

    def kd_build_ms(self):
        a = C.c_float(0)
        self._chk(self.L.sca_last_kd_build_ms(self.ctx, C.byref(a)), 'sca_last_kd_build_ms')
        return a.value

    def exchange_ms(self):
        """mean device time of the in-library all-gather over the profiled steps (0.0 without a communicator)"""
        a = C.c_float(0)
        self._chk(self.L.sca_last_exchange_ms(self.ctx, C.byref(a)), 'sca_last_exchange_ms')
        return a.value

    def pass_forms(self):
        """SCA_FORM_* bits of the last policy pass (which kernel forms the library picked)"""
        a = C.c_int(0)
        self._chk(self.L.sca_last_pass_forms(self.ctx, C.byref(a)), 'sca_last_pass_forms')
        return a.value

    def set_shard_emulation(self, on=True):
        self._chk(self.L.sca_set_shard_emulation(self.ctx, 1 if on else 0), 'sca_set_shard_emulation')

    def set_profiling(self, on=True):
        self._chk(self.L.sca_set_profiling(self.ctx, 1 if on else 0), 'sca_set_profiling')

    def agent_steps(self, reset=False):
        v = C.c_int64(0)
        self._chk(self.L.sca_agent_steps(self.ctx, C.byref(v), 1 if reset else 0), 'sca_agent_steps')
        return int(v.value)

    # ---- trajectory log (Agent.history_info) kept on the device -----------------------------------------
    def history_enable(self, capacity_rows):
        self._chk(self.L.sca_history_enable(self.ctx, int(capacity_rows)), 'sca_history_enable')

    def history_rows(self):
        a, b = C.c_int(0), C.c_int(0)
        self._chk(self.L.sca_history_rows(self.ctx, C.byref(a), C.byref(b)), 'sca_history_rows')
        return a.value, b.value

    def history(self, first_row=0, nrows=None, agent_begin=0, agent_count=None):
        """Rows [first_row, first_row+nrows) of agents [agent_begin, +agent_count): dict of [nrows, agents, 3] arrays."""
        if nrows is None:
            nrows = self.history_rows()[0] - first_row
        if agent_count is None:
            agent_count = self.n - agent_begin
        out = dict(pos=np.zeros((nrows, agent_count, 3)), heading=np.zeros((nrows, agent_count, 3)),
                   vel=np.zeros((nrows, agent_count, 3), np.float32))
        self._chk(self.L.sca_get_history(self.ctx, int(first_row), int(nrows), int(agent_begin), int(agent_count),
                                         _lib.ptr(out['pos'], C.c_double), _lib.ptr(out['heading'], C.c_double),
                                         _lib.ptr(out['vel'], C.c_float)), 'sca_get_history')
        return out

    # ---- multi-GPU --------------------------------------------------------------------------------------
    def set_shard(self, begin, count):
        self._chk(self.L.sca_set_shard(self.ctx, int(begin), int(count)), 'sca_set_shard')

    def public_records(self, which=0):
        p = C.c_void_p()
        b = C.c_int64()
        self._chk(self.L.sca_public_records(self.ctx, int(which), C.byref(p), C.byref(b)), 'sca_public_records')
        return p.value, b.value

    def bind_public_records(self, current_ptr, moved_ptr, bytes_each=0):
        self._chk(self.L.sca_bind_public_records(self.ctx, C.c_void_p(current_ptr), C.c_void_p(moved_ptr), int(bytes_each)),
                  'sca_bind_public_records')

    def step_begin(self, mode=NBR_KDTREE):
        self._chk(self.L.sca_step_begin(self.ctx, int(mode)), 'sca_step_begin')

    def step_end(self):
        self._chk(self.L.sca_step_end(self.ctx), 'sca_step_end')

    def set_stream(self, stream_ptr):
        """Run on this hipStream_t; 0 / None is HIP's null stream (torch's default stream), taken literally."""
        self._chk(self.L.sca_set_stream(self.ctx, C.c_void_p(stream_ptr)), 'sca_set_stream')

    def use_own_stream(self):
        self._chk(self.L.sca_use_own_stream(self.ctx), 'sca_use_own_stream')

    # ---- cell-owner partition of SCA_NBR_GRID with halo exchange (sca_partition_*) ---------------------------
    def partition_init(self, rank, nranks, axis=0, cuts=None, cap_halo=0, cap_mig=0):
        c = None if cuts is None else _lib.ptr(_lib.as_d(cuts), C.c_double)
        self._chk(self.L.sca_partition_init(self.ctx, int(rank), int(nranks), int(axis), c, int(cap_halo), int(cap_mig)), 'sca_partition_init')

    def partition_disable(self):
        self._chk(self.L.sca_partition_disable(self.ctx), 'sca_partition_disable')

    def partition_message_bytes(self):
        return int(self.L.sca_partition_message_bytes(self.ctx))

    def partition_pack(self, lower_ptr, upper_ptr):
        self._chk(self.L.sca_partition_pack(self.ctx, C.c_void_p(lower_ptr), C.c_void_p(upper_ptr)), 'sca_partition_pack')

    def partition_unpack(self, lower_ptr, upper_ptr):
        self._chk(self.L.sca_partition_unpack(self.ctx, C.c_void_p(lower_ptr), C.c_void_p(upper_ptr)), 'sca_partition_unpack')

    def partition_commit(self):
        self._chk(self.L.sca_partition_commit(self.ctx), 'sca_partition_commit')

    def partition_counts(self):
        a, b = C.c_int(0), C.c_int(0)
        self._chk(self.L.sca_partition_counts(self.ctx, C.byref(a), C.byref(b)), 'sca_partition_counts')
        return a.value, b.value

    def partition_owned(self):
        ids = np.zeros(self.n, np.int32)
        k = C.c_int(0)
        self._chk(self.L.sca_partition_owned(self.ctx, _lib.ptr(ids, C.c_int32), C.byref(k)), 'sca_partition_owned')
        return ids[:k.value].copy()

    # RCCL inside the library: run_steps then exchanges the shard's moved records itself, one host call per k steps
    def comm_probe(self):
        """True when the library can load RCCL (no collective: safe to call before the ranks agree on using it)."""
        return self.L.sca_comm_probe() == 0

    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        rc = self.L.sca_comm_unique_id(buf)
        if rc != 0:
            raise ScaError(f'sca_comm_unique_id: rc={rc} (librccl.so missing?)')
        return buf.raw

    def comm_init(self, rank, nranks, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._chk(self.L.sca_comm_init(self.ctx, int(rank), int(nranks), buf), 'sca_comm_init')

    def comm_destroy(self):
        self._chk(self.L.sca_comm_destroy(self.ctx), 'sca_comm_destroy')


def zaxis_flags(start, goal):
    """is_zAxis of scaPolicy.py:188-189: start and goal share x and y."""
    d = np.asarray(goal, float)[:, :3] - np.asarray(start, float)[:, :3]
    return ((np.abs(d[:, 0]) <= 1e-5) & (np.abs(d[:, 1]) <= 1e-5)).astype(np.uint8)
