/*
 * sca_hip.h -- C-ABI of libsca_hip.so: the MI355X-native batched velocity solver that replaces the
 * per-agent hot path of wuuya1/SCA (mamp/policies + neighbour search + the per-step loop in mamp/envs).
 *
 * Plain pointers and sizes only; caller-owned host buffers unless a function says "device".
 * Every function returns 0 on success or a negative sca_error; sca_last_error() gives the text.
 * One context per device; calls on one context are not thread-safe.
 *
 * What each entry point replaces in the reference (paths relative to wuuya1/SCA):
 *   sca_create / sca_set_agents   Agent.__init__ solver attributes            mamp/agents/agent.py:9-77
 *   sca_set_obstacles             Obstacle list + KDTree.buildObstacleTree    mamp/agents/obstacle.py:5-28, mamp/policies/kdTree.py:158-227
 *   sca_set_state / sca_get_state agent.pos/vel/heading/flags attribute reads  mamp/envs/mampenv.py:34-46
 *   sca_set_vpref                 SCA's Dubins-tracker output fed to intersect mamp/policies/sca/scaPolicy.py:32,264-338
 *   sca_policy_pass               first loop of MACAEnv._take_action:          mamp/envs/mampenv.py:28-40
 *                                 KDTree.buildAgentTree                        mamp/policies/kdTree.py:56-122
 *                                 computeNeighbors / insert*Neighbor           mamp/policies/sca/scaPolicy.py:107-116, mamp/agents/agent.py:79-124
 *                                 <Policy>.find_next_action + intersect        scaPolicy.py:26-240, rvo3dPolicy.py:23-179, srvo3dPolicy.py:23-231,
 *                                                                              orca3dPolicy.py:38-120,400-439, orca3dPolicyOfficial.py:37-300
 *   sca_env_update                second loop of _take_action + is_done        mamp/envs/mampenv.py:42-59,61-105
 *   sca_run_steps                 `while ...: env.step(actions)`               run_example/run_sca.py:174-178
 */
#ifndef SCA_HIP_H
#define SCA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCA_MAX_NEIGHBORS 16          /* agent.py:32 */
#define SCA_ACTION_DIM 7              /* mampenv.py:30: vx, vy, vz, speed, d_yaw, d_pitch, d_roll */
#define SCA_DIAG_DIM 5                /* n_suitable, fallback, chosen candidate, planeFail, lp4_ran (-1 = n/a) */

typedef struct sca_ctx sca_ctx;

/* agent.py:27-41 + config.py; sca_default_params() fills the reference's values.  The reference keeps these per Agent object; a context holds
 * ONE value of each as the default for all its agents; sca_set_agent_params hands over the agents' own where they differ
 * (sca_amd.env.MACAEnv reads them off the agents and does both).
 * sca_create refuses values the kernels were not built for: non-finite or non-positive distances / steps / speeds, max_neighbors outside
 * 1 .. 16, max_heading_change outside [0, pi].  Parity at non-default values: tests/golden/F16_params_*. */
typedef struct sca_params {
    double neighbor_dist;        /* agent.py:33  10.0 */
    double time_step;            /* agent.py:34  DT = 0.1 */
    double time_horizon;         /* agent.py:35  10.0 */
    double max_speed;            /* agent.py:36  1.0 */
    double max_heading_change;   /* agent.py:29  pi/4 */
    double near_goal_threshold;  /* config.py:3  0.5 */
    int32_t max_neighbors;       /* agent.py:32  16 (1 .. SCA_MAX_NEIGHBORS) */
    int32_t struct_bytes;        /* sizeof(sca_params) as the CALLER compiled it (sca_default_params_v2 / SCA_DEFAULT_PARAMS store it).
                                    0 = a version-100 caller: its struct ends here (56 bytes, this field was `reserved`, always 0), nothing
                                    beyond is read and dt_nominal = time_step, which is what version 100 integrated with */
    double dt_nominal;           /* agent.py:41  DT = 0.1: the integrator's step (mampenv.py:90-92); time_step is the one the constraints
                                    read (util.py:8, orca3dPolicyOfficial.py:98).  Version 101 on; read only when struct_bytes >= 64. */
} sca_params;

enum sca_policy {                 /* which find_next_action the agent runs */
    SCA_POLICY_SCA = 0,           /* mamp/policies/sca/scaPolicy.py        (v_pref supplied: sca_set_vpref) */
    SCA_POLICY_RVO3D = 1,         /* mamp/policies/rvo3dPolicy.py */
    SCA_POLICY_SRVO3D = 2,        /* mamp/policies/srvo3dPolicy.py */
    SCA_POLICY_ORCA3D = 3,        /* mamp/policies/orca3dPolicy.py         (sampled, as run_orca.py runs it) */
    SCA_POLICY_ORCA3D_LP = 4,     /* mamp/policies/orca3dPolicyOfficial.py (linearProgram1-4) */
    SCA_POLICY_RVO3D_DUBINS = 5   /* mamp/policies/sca/rvo3dDubinsPolicy.py (RVO3D selection, v_pref supplied) */
};

enum sca_flag { SCA_FLAG_AT_GOAL = 1, SCA_FLAG_COLLISION = 2, SCA_FLAG_TIMEOUT = 4 };   /* agent.py:70-72 */

enum sca_neighbor_mode {
    SCA_NBR_KDTREE = 0,           /* replica of the reference kd-tree (built and queried on the device): exact lists */
    SCA_NBR_GRID = 1,             /* uniform hashed grid, cells of neighbor_dist, rebuilt every step by a counting sort (three short
                                     launches instead of ~17 dependent tree levels; what multi-GPU runs want).  Lists hold the
                                     reference's (object, distSq) pairs whenever <= max_neighbors objects are in range; entries of
                                     equal distSq are ordered obstacles first, then by agent id (the reference: kd visit order);
                                     with more in range the 16 nearest are kept and SCA_ST_NBR_OVERFLOW is raised */
    SCA_NBR_KDTREE_HOSTBUILD = 2, /* same tree built on the host from a position read-back (debug / A-B reference) */
    SCA_NBR_AUTO = 3              /* the reference's lists, entry for entry, at the grid's price: the grid query for every agent, and the
                                     kd query (kdTree.py:124-156) for the agents whose list the grid cannot give exactly -- more than
                                     max_neighbors objects in range, or two objects of one kind at the same rounded distance (their order is
                                     the kd-tree's visit order, agent.py:87-90).  The kd-tree is still built every step -- its permutation is
                                     history (kdTree.py:43-45) -- but on a stream of its own beside the grid build and query, in
                                     sca_run_steps already behind the previous step's integrate stage, and a pass waits for it only when the
                                     grid query listed somebody (a device-side wait, hipStreamWaitValue32).  A pass is a plain SCA_NBR_KDTREE
                                     pass where that cannot pay: with the tracker inside the pass (the neighbour branch runs beside the
                                     re-plans there), and for 256 passes whenever the grid query listed more than an eighth of the shard
                                     (rings at the 16-neighbour density, lattices of identical cells).  Measured: random N = 4096 0.128 ->
                                     0.107 ms per step, N = 16 384 0.198 -> 0.160, N = 65 536 0.382 -> 0.333. */
};

enum sca_status_bit {             /* per-agent status word of the last policy pass */
    SCA_ST_SQRT_DOMAIN = 2,       /* reference would raise ValueError (scaPolicy.py:159); clamped here */
    SCA_ST_BAD_PREF_SPEED = 4,    /* np.arange(0.5, ps+0.03, ps-0.5) does not have 2 elements (scaPolicy.py:195) */
    SCA_ST_KD_STACK = 16,         /* kd traversal stack overflow (tree deeper than 48) */
    SCA_ST_NBR_OVERFLOW = 32,     /* grid mode: > max_neighbors in range, reference list is visit-order dependent */
    SCA_ST_VPREF_EDGE = 128,      /* reserved, never set.  Round 5 marked agent-steps whose straight-line v_pref (rvo3dPolicy.py:182-196)
                                     had a 5-decimal rounding within 1e-9 of flipping, because positions of a free-running episode
                                     carried ~1e-14 m of device sin / cos noise; since round 6 update_velocitie (mampenv.py:83-105) and
                                     cartesian2spherical (util.py:44-55) run on the restated glibc and the episode is the reference's */
    SCA_ST_TRACKER_EDGE = 64      /* reserved, never set.  Rounds 1-2 marked agent-steps whose device-tracker v_pref might differ from the
                                     reference's by a 5-decimal step (the device ran on another libm); since round 3 the device tracker
                                     computes glibc's bits (sca_amd/csrc/sca_glibc_math.h) and equals the host tracker bit for bit */
};

enum sca_error {
    SCA_OK = 0, SCA_ERR_ARG = -1, SCA_ERR_HIP = -2, SCA_ERR_STATE = -3, SCA_ERR_NOMEM = -4, SCA_ERR_UNSUPPORTED = -5
};

void sca_default_params(sca_params *p);                 /* the version-100 entry point: writes the first 56 bytes only (struct_bytes = 0) */
void sca_default_params_v2(sca_params *p, int32_t struct_bytes);   /* every field that fits into struct_bytes, and struct_bytes itself */
#define SCA_DEFAULT_PARAMS(p) sca_default_params_v2((p), (int32_t)sizeof(sca_params))
int sca_version(void);                     /* 102 (100: round 4; 101: dt_nominal appended to sca_params; 102: struct_bytes, sca_default_params_v2) */

int sca_create(const sca_params *p, int device, int max_agents, int max_obstacles, sca_ctx **out);
void sca_destroy(sca_ctx *ctx);
const char *sca_last_error(const sca_ctx *ctx);

/* static scene ---------------------------------------------------------------------------------- */
int sca_set_obstacles(sca_ctx *ctx, int m, const double *pos /*m*3*/, const double *radius /*m*/);
int sca_set_agents(sca_ctx *ctx, int n, const double *radius /*n*/, const double *pref_speed /*n*/,
                   const double *goal /*n*3*/, const uint8_t *policy /*n*/, const uint8_t *zaxis /*n*/,
                   const double *max_run_dist /*n*/);

/* The solver attributes PER AGENT, as the reference keeps them (agent.py:24-41: maxNeighbors, neighborDist, timeStep, timeHorizon, maxSpeed,
 * max_heading_change, dt_nominal are attributes of every Agent object, read by its own policy calls).  Arrays of n (= sca_set_agents' n); a NULL
 * array keeps the context's sca_params value for every agent; n = 0 or all NULL: back to one value per context.  Call after sca_set_agents
 * (which clears them) and before sca_device_tracker_enable.  The planner's two attributes, turning_radius and pitchlims, go per agent
 * through sca_device_tracker_set_agent_params (below).  Parity: tests/golden/F17_hetero_*. */
int sca_set_agent_params(sca_ctx *ctx, int n, const double *neighbor_dist, const int32_t *max_neighbors, const double *time_step,
                         const double *time_horizon, const double *max_speed, const double *max_heading_change, const double *dt_nominal);

/* dynamic state (host <-> device) ---------------------------------------------------------------- */
int sca_set_state(sca_ctx *ctx, const double *pos /*n*3*/, const float *vel /*n*3*/, const double *heading /*n*3*/,
                  const uint8_t *flags /*n*/, const double *total_dist /*n, nullable*/,
                  const int32_t *step_num /*n, nullable*/);
int sca_get_state(sca_ctx *ctx, double *pos, float *vel, double *heading, uint8_t *flags, double *total_dist,
                  int32_t *step_num); /* any pointer may be NULL */
/* kdTree.agentIDs: the permutation the reference carries from step to step (kdTree.py:43-45,101-111) */
int sca_set_kd_perm(sca_ctx *ctx, const int32_t *perm /*n*/);
int sca_get_kd_perm(sca_ctx *ctx, int32_t *perm /*n*/);
/* the agent kd-tree of the last policy pass, (2n-1) nodes x [begin,end,left,right,min3,max3] (kdTree.py:14-21) */
int sca_get_kd_tree(sca_ctx *ctx, double *tree_out /*(2n-1)*10*/);
/* externally computed preferred velocity (SCA / RVO3D+Dubins); mode[i]=1 uses vpref[i], 0 = straight line */
int sca_set_vpref(sca_ctx *ctx, const double *vpref /*n*3*/, const uint8_t *mode /*n*/);

/* the hot path ----------------------------------------------------------------------------------- */
int sca_policy_pass(sca_ctx *ctx, int neighbor_mode);
int sca_get_actions(sca_ctx *ctx, float *action /*n*7*/);
int sca_get_neighbors(sca_ctx *ctx, int32_t *nbr_n /*n*/, int32_t *nbr_id /*n*16*/, uint8_t *nbr_kind /*n*16*/,
                      double *nbr_dsq /*n*16*/, uint8_t *nbr_valid /*n*/);
/* distSq of agent.neighbors[0] after the last pass: >= 0 value, -1 empty list, -2 list not touched by the pass */
int sca_get_nbr0(sca_ctx *ctx, double *dsq0 /*n*/);
int sca_get_diag(sca_ctx *ctx, int32_t *diag /*n*5*/, int32_t *status /*n*/, double *vpref_used /*n*3*/);
int sca_env_update(sca_ctx *ctx, int *all_done /*nullable: skips the readback*/);
/* `steps` x (policy pass + env update) with the state resident in HBM; returns without synchronising.  steps == 0 does nothing;
 * steps < 0 is SCA_ERR_ARG, a neighbor_mode outside sca_neighbor_mode SCA_ERR_UNSUPPORTED, no state yet SCA_ERR_STATE (tests/test_gpu_abi_errors.py) */
int sca_run_steps(sca_ctx *ctx, int steps, int neighbor_mode);
/* MACAEnv.step (mampenv.py:22-25) in one call: one resident step (both loops of _take_action + is_done), then the number of agents of
 * this rank still running after it (0 == is_done) -- sca_run_steps(ctx, 1, mode) + sca_active_count with one stream synchronisation and
 * one pinned 32-KB read-back.  What `while not env.step()` of the drop-in env costs per step beyond the kernels (bench.py `env_api`).
 * It returns when the context's stream is through; an SCA_NBR_AUTO pass may still have its kd-tree build and the kd query of the listed
 * agents on the library's second stream -- the next sca_env_step copes with that as the steps inside sca_run_steps do, and EVERY other
 * entry point that takes the context first puts that stream in front of the context's (so whatever is read between steps is final). */
int sca_env_step(sca_ctx *ctx, int neighbor_mode, int *active);
int sca_synchronize(sca_ctx *ctx);
/* number of this rank's agents that are not done (at goal, collided or timed out) after the last env update; 0 == the
 * `all(agent.is_run_done)` of MACAEnv.is_done (mampenv.py:51-59).  Synchronises; reports a failed device kd build. */
int sca_active_count(sca_ctx *ctx, int *active);

/* multi-GPU / interop ------------------------------------------------------------------------------ */
/* This rank solves agents [begin, begin+count); all agents' public records must be present. */
int sca_set_shard(sca_ctx *ctx, int begin, int count);
/* Device address of a public-record array (48 B per agent: pos f64x3, vel f32x3, flags u32, radius f64).
 * which = 0: the current records; which = 1: the "moved" records written by sca_step_begin, i.e. the buffer an
 * RCCL all-gather (torch.distributed) exchanges between sca_step_begin and sca_step_end. */
int sca_public_records(sca_ctx *ctx, int which, void **device_ptr, int64_t *bytes_per_agent);
/* Use caller-owned device memory (e.g. two torch tensors of n*48 bytes, n of sca_set_agents; bytes_each says how large each
 * is and is checked) for the two record arrays; NULL, NULL restores the internal ones.  The n live records are carried over. */
int sca_bind_public_records(sca_ctx *ctx, void *current, void *moved, int64_t bytes_each /* >= n*48, n of sca_set_agents */);
/* One step split around the exchange: begin = kd build + neighbours + solve + integrate for this rank's shard
 * (writes the shard's moved records); [all-gather of the moved records]; end = collision / goal flags + publish. */
int sca_step_begin(sca_ctx *ctx, int neighbor_mode);
int sca_step_end(sca_ctx *ctx);
/* Run on a caller-provided hipStream_t, e.g. the torch stream a collective between sca_step_begin and sca_step_end is issued
 * on: the library's kernels and that collective are ordered only if they share the stream.  NULL is taken literally: HIP's
 * null stream (which is what torch's default stream is).  sca_use_own_stream() goes back to the context's own non-blocking
 * stream (the default after sca_create).  Both drain the stream in use first. */
int sca_set_stream(sca_ctx *ctx, void *hip_stream);
int sca_use_own_stream(sca_ctx *ctx);
/* RCCL inside the library (nothing in the reference: it is single-process).  One process per GPU; rank 0 calls
 * sca_comm_unique_id and hands the 128 bytes (an ncclUniqueId) to the other ranks by any means; every rank then calls
 * sca_comm_init after sca_set_agents.  From then on this rank owns agents [rank*n/nranks, (rank+1)*n/nranks) (n must divide),
 * (sca_set_shard / sca_set_shard_emulation return SCA_ERR_STATE while the communicator exists),
 * and every step of sca_run_steps is: shard's policy pass + integrate -> ncclAllGather of the shard's moved 48-byte records
 * on the library's stream -> collision / goal flags, i.e. a multi-GPU episode is ONE host call per k steps.  librccl.so is
 * loaded with dlopen at the first call; SCA_ERR_UNSUPPORTED when it is missing. */
/* 0 when librccl.so can be loaded and has every entry point the library uses, SCA_ERR_UNSUPPORTED otherwise.  No collective, no
 * device work: ranks call it and AGREE on the result before any of them calls sca_comm_init, which blocks inside
 * ncclCommInitRank until every rank has arrived. */
int sca_comm_probe(void);
int sca_comm_unique_id(void *id_out /*128 bytes*/);
int sca_comm_init(sca_ctx *ctx, int rank, int nranks, const void *unique_id /*128 bytes*/);
int sca_comm_destroy(sca_ctx *ctx);
/* Cell-owner partition of SCA_NBR_GRID with halo exchange (nothing in the reference: it is single-process).
 * Space is cut into slabs of grid cells along `axis` (0 x, 1 y, 2 z); rank r owns the agents whose cell lies in its slab and
 * holds copies of the agents in the one layer of cells on either side (the halo) -- everything the neighbour query
 * (kdTree.py:124-156, agent.py:79-99 on the grid) and the collision check (mampenv.py:61-80) of its agents look at.  Per step
 * it exchanges, with its two slab neighbours only, the old + moved records of the agents next to the cut and the private state
 * (heading, v_pref, distances, tracker record) of agents that crossed it, instead of all N records.  Results equal a single
 * rank's bit for bit (tests/test_gpu_partition.py).
 *   sca_partition_init   after sca_set_agents + sca_set_state with the COMPLETE state on every rank.  cuts: nranks - 1 ascending
 *                        coordinates along the axis, or NULL = equal shares of the agents as they stand; moved onto cell
 *                        boundaries.  cap_halo / cap_mig: entries per message (0 = n / 4, n / 16); an overflow is reported
 *                        by sca_partition_commit.  From then on the per-agent arrays of sca_get_* are meaningful for the owned
 *                        agents only (sca_partition_owned).  sca_set_state (complete again) re-derives the ownership.
 *   one step             sca_step_begin(SCA_NBR_GRID) -> sca_partition_pack into two DEVICE buffers of sca_partition_message_bytes()
 *                        (for the lower / the upper slab neighbour) -> exchange (what a rank packed for its lower neighbour is
 *                        what that neighbour unpacks as the message from ITS upper one) -> sca_partition_unpack(from lower,
 *                        from upper) -> sca_partition_commit (ownership moves; nothing waits for the device: the host sizes its
 *                        launches with bounds and learns the exact counts a step or two late) -> sca_step_end.
 *                        With one rank sca_run_steps does all of it. */
int sca_partition_init(sca_ctx *ctx, int rank, int nranks, int axis, const double *cuts /*nranks-1, nullable*/, int cap_halo, int cap_mig);
int sca_partition_disable(sca_ctx *ctx);
int64_t sca_partition_message_bytes(sca_ctx *ctx);
int sca_partition_pack(sca_ctx *ctx, void *device_buf_lower, void *device_buf_upper);       /* NULL where there is no neighbour */
int sca_partition_unpack(sca_ctx *ctx, const void *device_buf_lower, const void *device_buf_upper);
int sca_partition_commit(sca_ctx *ctx);
int sca_partition_counts(sca_ctx *ctx, int *owned, int *halo);
int sca_partition_owned(sca_ctx *ctx, int32_t *ids /*n*/, int *count);
/* average device time of the kernels of the last sca_policy_pass / sca_run_steps, measured with HIP events */
int sca_last_kernel_ms(sca_ctx *ctx, float *neighbors_ms, float *solve_ms, float *update_ms);

/* the same for the tracker's re-plan kernels (k_replan_group<4 .. 64 lanes per plan>, k_replan, k_track_replan), events on the stream they run on */
int sca_last_replan_ms(sca_ctx *ctx, float *replan_ms);
/* the kd build of kdTree.py:56-122 (k_kd_gather .. k_kd_block), events on the stream it ran on, every 16th build while profiling */
int sca_last_kd_build_ms(sca_ctx *ctx, float *kd_build_ms);
/* the same for the step's exchange when the library issues it (sca_comm_init: ncclAllGather inside sca_run_steps); 0 without a communicator */
int sca_last_exchange_ms(sca_ctx *ctx, float *exchange_ms);
/* Which kernel forms the last policy pass was launched with (the library picks them per pass from the shard size and the
 * re-plan count of a recent pass; none of them changes a result bit -- tests/test_gpu_solve_split.py, test_gpu_tracker.py):
 *   SCA_FORM_SOLVE_SPLIT   k_solve as k_solve_sweep (beside the tracker's re-plans) + k_solve_pick4 (behind them)
 *   SCA_FORM_TRACK_FUSED   k_track_replan instead of k_track + k_replan (with SCA_FORM_REPLAN_FEW: k_track_group, decision + 64-lane search per agent)
 *   SCA_FORM_REPLAN_LANE   the lane-per-plan re-plan kernel was launched (k_replan or k_track_replan)
 *   SCA_FORM_REPLAN_FEW    a k_replan_group kernel (4 .. 64 lanes per plan) was launched
 *   SCA_FORM_LP_LANE       the ORCA3D-Official agents went to k_lp (one lane per agent)
 *   SCA_FORM_SOLVE_FB      k_solve_fb: small shards solve and finish their fallbacks in one launch (no k_fallback launch)
 *   SCA_FORM_ACTION_FB     k_action_fb: shards of up to 16 384 agents run the fallback sweep inside the epilogue's launch (no k_fallback launch)
 *   SCA_FORM_AUTO_TAIL     SCA_NBR_AUTO: the kd query of the listed agents ran inside the pass's grid query (its last workgroup, from the tree the pass's
 *                          build publishes): no k_neighbors_kd_auto launch, no stream wait in front of the solve */
#define SCA_FORM_SOLVE_SPLIT 1
#define SCA_FORM_TRACK_FUSED 2
#define SCA_FORM_REPLAN_LANE 4
#define SCA_FORM_REPLAN_FEW 8
#define SCA_FORM_LP_LANE 16
#define SCA_FORM_SOLVE_FB 32
#define SCA_FORM_ACTION_FB 64
#define SCA_FORM_AUTO_TAIL 128
int sca_last_pass_forms(sca_ctx *ctx, int *forms);
/* SCA_NBR_AUTO statistics since the last reset: out4 = {AUTO passes, agents the grid query listed for the kd query (sum over the passes), the
 * largest list, passes in which somebody was listed}.  A pass with nobody listed never waits for the kd stream. */
int sca_auto_stats(sca_ctx *ctx, int64_t *out4, int reset);
/* Measurement aid for scaling models on one GPU: with a partial shard (sca_set_shard) and no communicator, sca_run_steps
 * runs what ONE rank of a larger job runs per step -- the replicated neighbour structure over all n agents, everything else
 * for the shard -- and copies the other agents' records over unchanged where the all-gather would deliver them. */
int sca_set_shard_emulation(sca_ctx *ctx, int on);

/* with profiling on, sca_run_steps brackets every kernel launch of the policy pass with HIP events on its stream;
 * sca_synchronize() then folds them into the averages sca_last_kernel_ms() returns */
int sca_set_profiling(sca_ctx *ctx, int on);
/* number of agents that entered find_next_action since the last reset (the metric's "agent-steps") */
int sca_agent_steps(sca_ctx *ctx, int64_t *count, int reset);

/* device self-test: numerators of l3norm(a_i, b_i) = round(|a_i - b_i|, 5) (mamp/util.py:104) as the solver's fast path
 * computes them (fast[]) and as the literal restatement does (exact[]); they must be identical */
int sca_selftest_l3norm(sca_ctx *ctx, int n, const double *a /*n*3*/, const double *b /*n*3*/, double *fast /*n*/, double *exact /*n*/);
/* The tracker's libm (sca_amd/csrc/sca_glibc_math.h: glibc 2.35's sin / cos / atan2 / acos / pow(x, 2) restated operation for
 * operation, so that the device computes the reference's -- i.e. Python's math module's -- bits).  fn: 0 sin(a), 1 cos(a),
 * 2 acos(a), 3 atan2(a, b), 4 pow(a, 2) -- the branch-free forms the kernels call; 5 sin, 6 cos, 7 atan2, 8 pow as the literal
 * restatements of glibc's control flow; 9 / 10 the sine / cosine of the fused sincos; 11 atan2, 12 sin, 13 cos, 14 pow(a, 2) as cartesian2spherical (util.py:44-55),
 * get_phi (util.py:145) and update_velocitie (mampenv.py:83-105) call them on the device (constant tables).  b may be NULL unless fn is 3, 7 or 11.  sca_selftest_libm evaluates on the device,
 * sca_selftest_libm_host on the host (no GPU needed); tests demand both equal the running glibc bit for bit. */
int sca_selftest_libm(sca_ctx *ctx, int fn, int n, const double *a, const double *b, double *out /*n*/);
int sca_selftest_libm_host(int fn, int n, const double *a, const double *b, double *out /*n*/);

/* Trajectory log = Agent.history_info (mamp/agents/agent.py:75-77, filled by to_vector :126-148 at the end of every
 * update_velocitie, mamp/envs/mampenv.py:105): one 64-byte row per agent per env step, kept in HBM so that resident
 * runs (sca_run_steps) need no per-step readback.  Row r = the r-th env step after sca_history_enable; every agent logs
 * every step, done agents included, as in the reference.  The goal and radius columns of ANIMATION_COLUMNS are constants
 * of sca_set_agents.  Steps beyond capacity_rows are counted as dropped, never overwritten.  With sca_set_shard a rank
 * logs its own shard only.  capacity_rows == 0 frees the log; sca_set_agents frees it too. */
int sca_history_enable(sca_ctx *ctx, int capacity_rows);
int sca_history_rows(sca_ctx *ctx, int *rows_logged, int *rows_dropped);
/* window [first_row, first_row+nrows) x [agent_begin, agent_begin+agent_count), row-major [row][agent][3]; any output
 * pointer may be null */
int sca_get_history(sca_ctx *ctx, int first_row, int nrows, int agent_begin, int agent_count, double *pos /*pos_x..z*/,
                    double *heading /*alpha, beta, gamma*/, float *vel /*vel_x..z*/);

/* native preferred-velocity tracker for SCAPolicy / RVO3dDubinsPolicy (host side, thread-parallel over agents):
 * replaces compute_v_pref / compute_dubins / update_dubins (mamp/policies/sca/scaPolicy.py:92-104,243-338) and the 3-D
 * Dubins planner (dubinsmaneuver3d.py:34-162, dubinsmaneuver2d.py:33-218,260-297).  Feed its output to sca_set_vpref. */
void *sca_tracker_create(int n, const double *goal /*n*3*/, const double *goal_heading /*n*3*/, const double *pref_speed /*n*/,
                         const uint8_t *zaxis /*n, nullable*/, double turning_radius /*agent.py:24 1.5*/,
                         double pitch_min, double pitch_max /*agent.py:27*/, double neighbor_dist /*agent.py:33*/);
/* agent.neighborDist per agent (scaPolicy.py:299 reads the agent's own when its list is empty); NULL: the one value of sca_tracker_create */
int sca_tracker_set_neighbor_dist(void *tracker, const double *neighbor_dist /*n, nullable*/);
/* agent.turning_radius / agent.pitchlims per agent for the host tracker (arrays of n, NULL = the constructor's value) */
int sca_tracker_set_agent_params(void *tracker, const double *turning_radius, const double *pitch_lo, const double *pitch_hi);
void sca_tracker_destroy(void *tracker);
/* one compute_v_pref per agent with active[i] != 0; nbr0_dsq[i] = distSq of agent.neighbors[0] as left by the previous
 * policy pass, negative when the list is empty (scaPolicy.py:299) */
int sca_tracker_vpref(void *tracker, const double *pos /*n*3*/, const float *vel /*n*3*/, const double *heading /*n*3*/,
                      const uint8_t *active /*n*/, const double *nbr0_dsq /*n*/, double *vpref_out /*n*3*/, int nthreads);
int sca_tracker_replans(void *tracker, int32_t *replans /*n*/);
/* dubinsmaneuver3d (dubinsmaneuver3d.py:34): q = [x, y, z, yaw, pitch]; samples = [x, y, z, psi, gamma] rows */
int sca_dubins_plan(const double *qi5, const double *qf5, double rmin, double pitch_min, double pitch_max, double *length,
                    char *mode7, int32_t *n_samples, double *samples /*nullable, cap*5*/, int cap);

/* What the tracker's bit-for-bit claim is conditional on.  Host and device tracker compute sin / cos / atan2 / acos / x ** 2 with a
 * restatement of ONE libm build -- GNU C Library 2.35, x86-64, the FMA variants (sca_amd/csrc/sca_glibc_math.h), the libm under the
 * Python that recorded tests/golden/.  math.sin & co. of a reference run on another host are THAT host's libm; if it is another
 * build the reference itself prints other last bits there, and this library keeps printing 2.35's.  sca_libm_check compares the
 * restatement with the running libm on a fixed set of 5 x 4096 arguments: returns 0 when they agree, 1 when they do not
 * (mismatches[5], nullable: sin, cos, atan2, acos, pow(x, 2)).  sca_tracker_create prints one note on stderr in the second case
 * (SCA_QUIET silences it), and so does the first sca_device_tracker_enable; sca_last_error stays reserved for failures (round 4 put the
 * note there after a SUCCESSFUL enable; callers that test it for emptiness saw a failure).  Thread-safe (one check per process).
 * Behaviour never changes. */
int sca_libm_check(int64_t *mismatches);

/* The same tracker on the device: one lane per agent for the tracking, tracker records resident in HBM, the re-planning
 * agents of a step compacted into kernels of their own -- one lane up to one wavefront per plan, by how many a step has
 * (sca_amd/csrc/sca_tracker.hip.h).  Same statements as the host tracker AND the same libm (sca_glibc_math.h: glibc's sin / cos /
 * atan2 / acos / pow restated operation for operation, device build checked against the host's bit for bit): v_pref, every
 * follow-or-re-plan decision and every plan equal the host tracker's -- i.e. the reference's -- bit for bit
 * (tests/test_gpu_tracker.py).
 * sca_device_tracker_enable: agents with policy SCA / RVO3D_DUBINS take v_pref from it from now on -- inside every
 * sca_policy_pass / sca_step_begin / sca_run_steps when in_pass != 0 (agent.neighbors[0] of the previous pass is read from
 * the neighbour lists on the device), otherwise only through sca_device_tracker_vpref.  sca_set_agents disables it. */
int sca_device_tracker_enable(sca_ctx *ctx, const double *goal_heading /*n*3, agent.py:19*/, double turning_radius,
                              double pitch_min, double pitch_max, int in_pass);
/* the tracked agents take v_pref from their policy's own straight-line rule again (sca_set_vpref afterwards to feed it from the host) */
int sca_device_tracker_disable(sca_ctx *ctx);
/* agent.turning_radius / agent.pitchlims PER AGENT (the reference keeps them on every Agent object; scaPolicy.py:95,272,302 read the agent's own).
 * Arrays of n, a NULL array = sca_device_tracker_enable's value for everybody, all NULL = back to that one value.  Entries of untracked agents
 * (policy not SCA / RVO3D_DUBINS) are ignored; a tracked agent needs turning_radius > 0 and pitch_lo < pitch_hi (SCA_ERR_ARG otherwise).
 * Up to 16 distinct (turning_radius, pitch_lo, pitch_hi) among the tracked agents: classes -- the re-plan kernels run once per class with the
 * class's values as kernel arguments (the search keeps the three in scalar registers).  More (the reference has no limit): the per-agent
 * form -- every re-plan gets a wavefront of its own, which loads its agent's values; slower for large re-plan counts, never refused.
 * After sca_device_tracker_enable.  Parity: tests/golden/F18_hetero_track_* (3 x 3 classes; F18_hetero_track_circle30_each: 30 settings). */
int sca_device_tracker_set_agent_params(sca_ctx *ctx, int n, const double *turning_radius, const double *pitch_lo, const double *pitch_hi);
/* one compute_v_pref per active tracked agent on the current state; nbr0_dsq as in sca_tracker_vpref, NULL = from the
 * device's neighbour lists; vpref_out nullable */
int sca_device_tracker_vpref(sca_ctx *ctx, const double *nbr0_dsq /*n, nullable*/, double *vpref_out /*n*3, nullable*/);
int sca_device_tracker_replans(sca_ctx *ctx, int32_t *replans /*n*/);

/* diagnostics: the tracker record of one agent as 24 doubles -- horizontal maneuver (r_min, t, p, length), vertical maneuver
 * (the same four), plan length, sampling size, rounds of the speculative search that produced the plan (0: a one-step-at-a-time form), one
 * unused slot, cursor, sample count, tracked node[3], untruncated v_pref[3],
 * the two words, 64 x candidate radii tried, re-plan count -- of the host tracker / the device tracker */
int sca_tracker_debug(void *tracker, int agent, double *out24);
int sca_device_tracker_debug(sca_ctx *ctx, int agent, double *out24);
/* host self-test (no GPU needed): the device planner's four-lane form evaluates the four CSC Dubins words
 * (dubinsmaneuver2d.py:33-109) as one sign-parametrised instruction stream; this compares it with the literal words on the
 * given frames (alpha, beta in [0, 2 pi), d >= 0) and counts results that are not bit-identical (must be 0) */
int sca_selftest_dubins_words(int n, const double *alpha, const double *beta, const double *d, int64_t *mismatches);
/* host self-test (no GPU needed): the device's lane-per-plan kernels run the 3-D planner's search in a lean form (a candidate
 * radius evaluated for feasibility and length only; far problems through a straight-line block; sca_dubins.hpp, plan3d_lean);
 * this runs that form, compiled for the host, and the literal planner (dubinsmaneuver3d.py:34-113) on n poses q[n][10] =
 * (qi[5], qf[5]) and counts plans that are not bit-identical (must be 0); the other two outputs (nullable) say how many
 * candidates took the lean block and how many the literal construction */
int sca_selftest_plan3d_lean(int n, const double *q, double turning_radius, double pitch_lo, double pitch_hi, int64_t *mismatches,
                             int64_t *lean_candidates, int64_t *literal_candidates);

/* host-only helpers (no GPU needed) ----------------------------------------------------------------- */
/* unit Fibonacci directions of scaPolicy.py:195-200 (SoA [3][num_N]) and the get_phi numerators */
int sca_candidate_table(int num_N, double *unit /*3*num_N*/, double *phi_num /*num_N*/);
/* replica of KDTree.buildAgentTreeRecursive: permutes perm in place, writes (2n-1) nodes of 10 doubles
 * [begin,end,left,right,min3,max3] when tree_out != NULL */
int sca_kd_build_host(int n, const double *pos /*n*3*/, int32_t *perm /*n*/, double *tree_out);

#ifdef __cplusplus
}
#endif
#endif
