"""Drop-in host side for mamp.envs / mamp.agents / mamp.policies of wuuya1/SCA, backed by libsca_hip.

Same names, arguments and attribute surface as the reference (file:line cites are into wuuya1/SCA):
    Agent(start_pos, goal_pos, vel, radius, pref_speed, policy, id, dt)      mamp/agents/agent.py:9
    Obstacle(pos, shape_dict, id)                                            mamp/agents/obstacle.py:5
    SCAPolicy / RVO3DPolicy / SRVO3DPolicy / ORCA3DPolicy / ORCA3DPolicyOfficial / RVO3dDubinsPolicy
        .find_next_action(dict_comm, agent, kdTree) -> action[7]             e.g. mamp/policies/rvo3dPolicy.py:23
    MACAEnv().set_agents(agents, obstacles=[...]); MACAEnv.step(actions) -> bool   mamp/envs/mampenv.py:16-25

Differences that matter:
  * the env makes ONE library call per step for all agents (the reference calls find_next_action per agent,
    mampenv.py:34-40); `agent.policy.find_next_action(...)` still works for a single agent -- it runs the same
    batched pass and returns that agent's row;
  * per-agent state lives in structure-of-arrays owned by the env; the Agent attributes the reference exposes
    (pos_global_frame, vel_global_frame, heading_global_frame, is_at_goal, ...) are views / properties over them;
  * SCAPolicy and RVO3dDubinsPolicy take v_pref from the reference's Dubins tracker (scaPolicy.py:264-338), a host-side,
    per-agent stateful planner outside the kernel boundary.  The native restatement is sca_amd.tracker.DubinsTracker:
    pass it as MACAEnv(v_pref_fn=tracker) (any `v_pref_fn(env) -> [N,3]` works; bit-exact, host-bound), or use
    MACAEnv(device_tracker=True): the same tracker as kernels inside every pass (state stays in HBM; equal to the host
    tracker bit for bit -- same statements, same restated glibc libm); without either the straight-line rule of rvo3dPolicy.py:182-196
    is used and `env.dubins_tracker` is False;
  * the solver attributes of agent.py:24-41 -- turning_radius and pitchlims included -- are read off the agents in set_agents and travel per
    agent where the agents differ;
  * history logging (agent.py:126-147, pandas) and the per-step prints are not reproduced.
There is no CPU path: constructing the env without a GPU raises.
"""
import math

import numpy as np

from . import scenarios as _sc
from . import solver as S

DT = 0.1                      # mamp/configs/config.py:2
NEAR_GOAL_THRESHOLD = 0.5     # mamp/configs/config.py:3


class _Policy:
    policy_id = None
    needs_external_vpref = False

    def __init__(self):                      # constructible with no arguments (agent.py:11 `policy()`)
        self.now_goal = None
        self.type = 'internal'
        self._env = None

    def find_next_action(self, dict_comm, agent, kdTree):
        """Single-agent entry point of the reference API.  Runs the batched policy pass of the owning env (cached per
        step) and returns this agent's [vx, vy, vz, speed, d_yaw, d_pitch, d_roll]."""
        if self._env is None:
            raise RuntimeError('policy is not attached to a MACAEnv (call env.set_agents first)')
        self.now_goal = agent.goal_global_frame          # get_trajectory: agent.path is always empty (scaPolicy.py:89)
        return self._env._policy_row(agent.id)


class SCAPolicy(_Policy):
    policy_id = S.POL_SCA
    needs_external_vpref = True


class RVO3DPolicy(_Policy):
    policy_id = S.POL_RVO3D


class SRVO3DPolicy(_Policy):
    policy_id = S.POL_SRVO3D


class ORCA3DPolicy(_Policy):                 # mamp/policies/orca3dPolicy.py (sampled, the one run_orca.py imports)
    policy_id = S.POL_ORCA3D


class ORCA3DPolicyOfficial(_Policy):         # mamp/policies/orca3dPolicyOfficial.py (linearProgram1-4)
    policy_id = S.POL_ORCA3D_LP


class RVO3dDubinsPolicy(_Policy):
    policy_id = S.POL_RVO3D_DUBINS
    needs_external_vpref = True


class Obstacle:
    def __init__(self, pos, shape_dict, id):
        self.shape = shape_dict['shape']
        self.feature = shape_dict['feature']
        if self.shape == 'cube':                                    # obstacle.py:9-11: bounding sphere
            self.length, self.width, self.height = shape_dict['feature']
            self.radius = math.sqrt(self.length ** 2 + self.width ** 2 + self.height ** 2) / 2
        elif self.shape == 'sphere':
            self.radius = shape_dict['feature']
        else:
            raise NotImplementedError
        self.pos_global_frame = np.array(pos, dtype='float64')
        self.vel_global_frame = np.array([0.0, 0.0, 0.0])
        self.pos = pos
        self.id = id
        self.is_at_goal = True
        self.is_obstacle = True
        self.is_collision = False


class Agent:
    def __init__(self, start_pos, goal_pos, vel, radius, pref_speed, policy, id, dt=0.1):
        self.group = 0
        self.policy = policy()
        self.initial_pos = np.array(start_pos, dtype='float64')
        self.goal_pos = np.array(goal_pos, dtype='float64')
        self._pos = np.array(start_pos[:3], dtype='float64')
        self.goal_global_frame = np.array(goal_pos[:3], dtype='float64')
        self._heading = np.array(start_pos[3:], dtype='float64')
        self.goal_heading_frame = np.array(goal_pos[3:], dtype='float64')
        self.initial_heading = np.array(start_pos[3:], dtype='float64')
        self._vel = np.array(vel, dtype=np.float32)
        self.radius = radius
        self.turning_radius = 1.5
        self.id = id
        self.pref_speed = pref_speed
        self.pitchlims = [-math.pi / 4, math.pi / 4]
        self.min_heading_change = self.pitchlims[0]
        self.max_heading_change = self.pitchlims[1]
        self.maxNeighbors = 16
        self.neighborDist = 10.0
        self.timeStep = DT
        self.timeHorizon = 10.0
        self.maxSpeed = 1.0
        self.dt_nominal = DT
        self.path = []
        self._v_pref = np.zeros(3)
        self.is_obstacle = False
        d = float(np.sqrt(((self.initial_pos[:3] - self.goal_pos[:3]) ** 2).sum()))
        self.straight_path_length = round(d, 5) - 0.5               # agent.py:51
        self.desire_steps = int(self.straight_path_length / (pref_speed * DT))
        self.max_run_dist = 3.0 * round(d, 5)                       # agent.py:74
        self._env = None
        self._flags = 0
        self._total_dist = 0.0
        self._step_num = 0

    # --- state: views over the env's arrays once attached -------------------------------------------------------------
    def _get(self, name, local):
        return local if self._env is None else getattr(self._env, name)[self.id]

    # The three attributes the reference's callers read per agent per step (run_sca.py:181-196).  The env's host mirrors are refreshed IN
    # PLACE (one bulk read-back on first use after a step), so an agent's row view is made once and stays valid: a read is an attribute
    # test, a staleness test and a return (100 000 reads per step went through three property hops and a fresh view).
    _row_pos = _row_vel = _row_heading = None

    @property
    def pos_global_frame(self):
        env = self._env
        if env is None:
            return self._pos
        if env._stale:
            env._state('pos')
        r = self._row_pos
        if r is None:
            r = self._row_pos = env._mirror['pos'][self.id]
        return r

    @property
    def vel_global_frame(self):
        env = self._env
        if env is None:
            return self._vel
        if env._stale:
            env._state('vel')
        r = self._row_vel
        if r is None:
            r = self._row_vel = env._mirror['vel'][self.id]
        return r

    @property
    def heading_global_frame(self):
        env = self._env
        if env is None:
            return self._heading
        if env._stale:
            env._state('heading')
        r = self._row_heading
        if r is None:
            r = self._row_heading = env._mirror['heading'][self.id]
        return r

    @property
    def total_time(self):
        """agent.total_time (scaPolicy.py:30,35-37,62-64): wall time this agent's find_next_action calls took.  The batched
        pass serves every active agent at once, so a step's wall time is shared equally among the agents it served."""
        if self._env is None:
            return 0.0
        cum = self._env._time_cum
        return cum[min(self.step_num, len(cum) - 1)]

    @property
    def v_pref(self):
        """agent.v_pref as the last find_next_action left it (scaPolicy.py:337, rvo3dPolicy.py:195)"""
        if self._env is None:
            return self._v_pref
        return self._env._vpref_of(self.id)

    @property
    def total_dist(self):
        return float(self._get('total_dist', self._total_dist))

    @property
    def step_num(self):
        return int(self._get('step_num', self._step_num))

    def _flag(self, bit):
        f = self._flags if self._env is None else int(self._env.flags[self.id])
        return bool(f & bit)

    @property
    def is_at_goal(self):
        return self._flag(S.FLAG_AT_GOAL)

    @property
    def is_collision(self):
        return self._flag(S.FLAG_COLLISION)

    @property
    def is_out_of_max_time(self):
        return self._flag(S.FLAG_TIMEOUT)

    @property
    def is_run_done(self):
        return self.is_at_goal or self.is_collision or self.is_out_of_max_time

    @property
    def neighbors(self):
        """[(object, distSq)] of the last policy pass, as agent.py:79-124 leaves it."""
        if self._env is None:
            return []
        return self._env._neighbors_of(self.id)


class _KdTreeView:
    """Stand-in for mamp.policies.kdTree.KDTree: the tree itself lives on the device; the permutation the reference
    carries from step to step (kdTree.py:43-45) is readable."""

    def __init__(self, env):
        self._env = env
        self.max_leaf_size = 10

    @property
    def agentIDs(self):
        return list(self._env.solver.get_kd_perm())


class MACAEnv:
    def __init__(self, v_pref_fn=None, device=0, neighbor_mode=S.NBR_KDTREE, history_capacity=0, device_tracker=False):
        self.agents = None
        self.obstacles = []
        self.kdTree = None
        self.solver = None
        self.v_pref_fn = v_pref_fn
        self.device_tracker = bool(device_tracker)     # SCA / RVO3D+Dubins v_pref from the tracker kernels inside every pass
        if self.device_tracker and v_pref_fn is not None:
            raise ValueError('device_tracker=True and v_pref_fn are alternatives')
        self.dubins_tracker = v_pref_fn is not None or self.device_tracker
        self.device = device
        self.neighbor_mode = neighbor_mode
        self.history_capacity = history_capacity      # env steps of Agent.history_info kept on the device (0: no log);
        self.history_budget_bytes = 64 << 30          # 64 B per agent per step, allocated up front: capped to this budget
        self._row_cache = None
        self._last_neighbors = None     # neighbour lists of the previous policy pass (read by the v_pref tracker)
        self._time_cum = [0.0]          # [s] = seconds of policy wall time per served agent over the first s env steps
        self._active = 0                # agents the next step will serve
        self._vpref_cache = None

    def set_agents(self, agents, obstacles=None):
        if obstacles is None:
            raise TypeError('obstacles must be a list (the reference crashes on None too, kdTree.py:48)')
        for i, a in enumerate(agents):
            if a.id != i:
                raise ValueError('agent.id must equal its list index (kdTree.py:64)')
        self.agents = agents
        self.obstacles = obstacles
        n, m = len(agents), len(obstacles)
        self._mirror = dict(pos=np.array([a._pos for a in agents], dtype=np.float64).reshape(n, 3),
                            vel=np.array([a._vel for a in agents], dtype=np.float32).reshape(n, 3),
                            heading=np.array([a._heading for a in agents], dtype=np.float64).reshape(n, 3),
                            flags=np.zeros(n, np.uint8), total_dist=np.zeros(n), step_num=np.zeros(n, np.int32))
        self._stale = False
        self.goal = np.array([a.goal_global_frame for a in agents], dtype=np.float64).reshape(n, 3)
        self.policy_ids = np.array([a.policy.policy_id for a in agents], np.uint8)
        self._ext = np.array([a.policy.needs_external_vpref for a in agents], bool)
        start = np.array([a.initial_pos for a in agents], dtype=np.float64)
        goal6 = np.array([a.goal_pos for a in agents], dtype=np.float64)
        # the solver attributes the reference keeps per Agent (agent.py:24-41) and its policies read per call (scaPolicy.py:95,112,272,302,
        # util.py:8,17, orca3dPolicyOfficial.py:44,98,108, agent.py:87-99, mampenv.py:90-92): a context holds one value of each
        # Attributes all agents agree on become the context's sca_params; the ones that differ from agent to agent go to the device as
        # per-agent arrays (sca_set_agent_params), the context keeping the first agent's value as its default.
        attr_of = dict(neighbor_dist=('neighborDist', float), max_neighbors=('maxNeighbors', int), time_step=('timeStep', float),
                       time_horizon=('timeHorizon', float), max_speed=('maxSpeed', float), max_heading_change=('max_heading_change', float),
                       dt_nominal=('dt_nominal', float))
        params, per_agent = {}, {}
        for name, (attr, conv) in attr_of.items():
            vals = [conv(getattr(a, attr)) for a in agents]
            params[name] = vals[0]
            if any(v != vals[0] for v in vals):
                per_agent[name] = vals
        self.solver = S.BatchedSolver(max_agents=n, max_obstacles=max(m, 1), device=self.device, params=params)
        self.solver.set_obstacles(np.array([o.pos_global_frame for o in obstacles], dtype=np.float64).reshape(m, 3),
                                  np.array([o.radius for o in obstacles], dtype=np.float64))
        self.solver.set_agents([a.radius for a in agents], [a.pref_speed for a in agents], self.goal, self.policy_ids,
                               S.zaxis_flags(start, goal6), [a.max_run_dist for a in agents])
        if per_agent:
            self.solver.set_agent_params(**per_agent)
        self.per_agent_attributes = sorted(per_agent)             # which attributes the agents disagree on (empty: one value per context)
        self.solver.set_state(self.pos, self.vel, self.heading, self.flags)
        if self.device_tracker and self._ext.any():
            # the planner's attributes (agent.turning_radius, agent.pitchlims: scaPolicy.py:95,272,302): the first tracked agent's become the
            # tracker's defaults, and where the tracked agents differ every agent's own go over (classes of equal values on the device)
            tracked = [a for a in agents if a.policy.needs_external_vpref]
            trip = [(float(a.turning_radius), float(a.pitchlims[0]), float(a.pitchlims[1])) for a in agents]
            first = (float(tracked[0].turning_radius), float(tracked[0].pitchlims[0]), float(tracked[0].pitchlims[1]))
            self.solver.device_tracker_enable(goal6[:, 3:6], turning_radius=first[0], pitchlims=(first[1], first[2]))
            if any((float(a.turning_radius), float(a.pitchlims[0]), float(a.pitchlims[1])) != first for a in tracked):
                self.solver.device_tracker_set_agent_params([t[0] for t in trip], [t[1] for t in trip], [t[2] for t in trip])
                self.per_agent_attributes = sorted(self.per_agent_attributes + ['turning_radius / pitchlims'])
        for a in agents:
            a._env = self
            a._row_pos = a._row_vel = a._row_heading = None        # (row views belong to the mirrors of the env they were made for)
            a.policy._env = self
        self.kdTree = _KdTreeView(self)
        self._row_cache = None
        self._nbr_cache = None
        self._time_cum = [0.0]
        self._active = n
        self._vpref_cache = None
        # the log costs 64 B x rows x agents of HBM up front: cap it to a budget instead of failing in hipMalloc
        if self.history_capacity and 64 * self.history_capacity * n > self.history_budget_bytes:
            capped = max(1, self.history_budget_bytes // (64 * n))
            import warnings
            warnings.warn(f'history_capacity {self.history_capacity} x {n} agents x 64 B exceeds the {self.history_budget_bytes >> 30} GiB '
                          f'budget: keeping the first {capped} env steps (later steps are counted as dropped)')
            self.history_capacity = capped
        if self.history_capacity:
            self.solver.history_enable(self.history_capacity)

    # ---- host mirrors of the device state, refreshed on first use after a step (the reference's per-agent attributes) -------
    def _state(self, name):
        if self._stale:
            st = self.solver.get_state()
            for k in self._mirror:
                self._mirror[k][...] = st[k]
            self._stale = False
        return self._mirror[name]

    pos = property(lambda self: self._state('pos'))
    vel = property(lambda self: self._state('vel'))
    heading = property(lambda self: self._state('heading'))
    flags = property(lambda self: self._state('flags'))
    total_dist = property(lambda self: self._state('total_dist'))
    step_num = property(lambda self: self._state('step_num'))

    # ---- one step = MACAEnv.step (mampenv.py:22-25) ---------------------------------------------------------------------
    def _policy_pass(self):
        if self._row_cache is None:
            if self.v_pref_fn is not None and self._ext.any():
                vp = np.asarray(self.v_pref_fn(self), dtype=np.float64).reshape(len(self.agents), 3)
                self.solver.set_vpref(vp, self._ext.astype(np.uint8))
            self.solver.policy_pass(self.neighbor_mode)
            self._row_cache = self.solver.actions()
            self._nbr_cache = None
            if self.v_pref_fn is not None:
                self._nbr_cache = self._last_neighbors = self.solver.neighbors()
            self._stale = True                                    # is_collision set inside insert*Neighbor (agent.py:84)
        return self._row_cache

    @property
    def all_actions(self):
        """The [N, 7] float32 action rows of the last step (the reference's `all_actions` inside _take_action, mampenv.py:31,40): read back
        from the device when asked for."""
        if self._row_cache is not None:
            return self._row_cache
        return self.solver.actions()

    def _policy_row(self, i):
        return list(self._policy_pass()[i])

    def _vpref_of(self, i):
        if self._vpref_cache is None:
            self._vpref_cache = np.nan_to_num(self.solver.diag()['vpref'])
        return self._vpref_cache[i]

    def _neighbors_of(self, i):
        if self._nbr_cache is None:
            self._nbr_cache = self.solver.neighbors()
        nb = self._nbr_cache
        out = []
        for k in range(int(nb['nbr_n'][i])):
            j = int(nb['nbr_id'][i, k])
            obj = self.obstacles[j] if nb['nbr_kind'][i, k] else self.agents[j]
            out.append((obj, float(nb['nbr_dsq'][i, k])))
        return out

    def step(self, actions=None):
        """`actions` is ignored, as in the reference (mampenv.py:22).  Without a host-side v_pref_fn the whole step (both loops
        of _take_action, mampenv.py:27-49) is one resident library call and nothing but the done count comes back: the
        per-agent attributes are read from the device the next time somebody looks at them."""
        import time
        t0 = time.perf_counter()
        served = max(1, self._active)
        if self._row_cache is None and self.v_pref_fn is None:
            self._active = self.solver.env_step(self.neighbor_mode)   # one call; synchronises: the step is over when this returns
            t_policy = time.perf_counter() - t0
            done = self._active == 0
            self._nbr_cache = None
        else:
            self._policy_pass()
            t_policy = time.perf_counter() - t0                  # the reference times find_next_action only
            done = self.solver.env_update()
            self._active = self.solver.active_count()
        self._time_cum.append(self._time_cum[-1] + t_policy / served)
        self._stale = True
        self._row_cache = None
        self._vpref_cache = None
        return done


def build_circle_agents(n, policy=RVO3DPolicy, rad=None, radius=0.5, pref_speed=1.0):
    """run_example/run_sca.py:106-126 build_agents for the circle scenario."""
    sc = _sc.circle(n, rad=rad)
    return [Agent(start_pos=list(sc['start'][i]), goal_pos=list(sc['goal'][i]), vel=[0.0, 0.0, 0.0], radius=radius,
                  pref_speed=pref_speed, policy=policy, id=i, dt=DT) for i in range(n)]
