// sca_kernels.hip.h -- gfx950 kernels of the batched SCA / RVO3D / S-RVO3D / ORCA3D velocity solver.
//
//   k_neighbors_kd / k_neighbors_kd4 : replica of KDTree.query*TreeRecursive + Agent.insert*Neighbor (kdTree.py:127-262,
//                    agent.py:79-124): one wavefront per agent, or four agents per wavefront (16 lanes = one DPP row
//                    each) once the shard fills the chip; bounded sorted list one entry per lane.
//   k_solve        : cone / half-space construction + posture filter + compacted candidate sweep + selection (or
//                    LP1-4), ONE WAVEFRONT (64 lanes) PER AGENT, neighbour constants staged in LDS and read as
//                    wave-uniform broadcasts, selection by packed integer keys and DPP wave minima.
//                    k_solve_full finishes the rare agents without a suitable candidate (compute_without_suitV).
//   k_action       : cartesian2spherical + float32 action row, one lane per agent; also update_velocitie
//   k_fallback     : the agents without a suitable candidate (complete 513-candidate sweep, one wavefront each) + their epilogue
//                    (mampenv.py:83-105) and the trajectory log when the state stays resident (sca_run_steps).
//   k_collide_finish : check_agent_state + is_done (mampenv.py:51-80), eight agents per wavefront.
//
// HBM layout: one 48-byte PubRec per agent (pos f64x3, vel f32x3, flags, radius) -- the only thing other
// agents / other GPUs read; private per-agent arrays are SoA.  No MFMA: there is no contraction here.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>

#include "sca_core.h"

namespace sca {

struct KdNode { int begin, end, left, right; double mn[3], mx[3]; };   // 64 B, kdTree.py:14-21
// What the query reads: the node header plus BOTH children's boxes, so that a level of the descent costs one
// (scalar) load instead of two dependent ones.  128 B.
// Query record of a tree node: header + BOTH children's boxes, interleaved so that the packed neighbour kernel can add
// the six box terms of the left child on lane 2 and of the right child on lane 3 with the same lane shifts:
// bx = mn0L mn0R mx0L mx0R mn1L mn1R mx1L mx1R mn2L mn2R mx2L mx2R   (index (2*axis + is_max)*2 + side)
struct alignas(16) KdWide { int begin, end, left, right; double bx[12]; double pad[2]; };
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int kdw_idx(int side, int is_max, int axis) { return (2 * axis + is_max) * 2 + side; }
static_assert(sizeof(KdWide) == 128, "KdWide must be 128 bytes");
struct ObsRec { double px, py, pz, radius; };                          // obstacle.py:5-28 (sphere)

// One row of the trajectory log: what Agent.to_vector (agent.py:126-148) appends to history_info at the end of
// update_velocitie (mampenv.py:105); the goal and radius columns are per-agent constants and stay on the host.
struct alignas(16) HistRow {
    double px, py, pz;       // pos_x, pos_y, pos_z
    double a, b, g;          // alpha, beta, gamma
    float vx, vy, vz;        // vel_x, vel_y, vel_z
    uint32_t flags;          // flags the agent entered the step with (not a reference column)
};
static_assert(sizeof(HistRow) == 64, "HistRow must be 64 bytes");

struct DeviceView {
    // public state
    PubRec *rec;             // [n]
    PubRec *rec_new;         // [n] scratch for the integrate/collide split
    // private per-agent state (SoA)
    double *heading;         // [n*3]
    double *goal;            // [n*3]
    double *pref_speed;      // [n]
    double *vpref_ext;       // [n*3]
    double *total_dist;      // [n]
    double *max_run_dist;    // [n]
    int32_t *step_num;       // [n]
    uint8_t *vpref_mode;     // [n]
    uint8_t *policy;         // [n]
    uint8_t *zaxis;          // [n]
    // obstacles + trees
    ObsRec *obs;             // [m]
    ObsRec *obs_sorted;      // [m] obstacles in obstacle-tree position order (leaf members are contiguous)
    KdWide *awide;           // [2n] query view of the agent tree
    KdWide *owide;           // [2m] query view of the obstacle tree
    double *kx, *ky, *kz;    // [n] agent coordinates in agent-tree position order (written by the build)
    KdNode *atree;           // [2n]
    int32_t *aperm;          // [n]
    KdNode *otree;           // [2m]
    int32_t *operm;          // [m]
    // neighbour lists
    int32_t *nbr_n;          // [n]
    int32_t *nbr_id;         // [n*16]  (obstacles carry NBR_OBSTACLE_BIT)
    double *nbr_dsq;         // [n*16]
    uint32_t *coll_new;      // [n] collision detected by insert*Neighbor in this pass
    uint8_t *nbr_valid;      // [n]
    int32_t *near_n;         // [n] collision candidates found by K1 (-1: not computed / overflow -> traversal)
    int32_t *near_id;        // [n*NEAR_MAX] ids within collision reach at the step's old positions
    // outputs
    float *action;           // [n*8] (7 used)
    double *vpref_used;      // [n*3]
    double *vpost;           // [n*3] selected velocity (k_solve -> k_action)
    void *prep;              // [n] Prep records (per-agent scalar prologue)
    int32_t *fb_list;        // [n] agents without a suitable candidate: finished by k_solve_full
    int32_t *fb_count;       // [1]
    uint8_t *is_fb;          // [n] 0: vpost holds the new velocity; 1: k_solve handed the agent to the fallback list (the epilogue
                             //     kernel's second half finishes it); 2: the epilogue derives vpost from the chosen candidate index
    int32_t *diag;           // [n*8]: n_suit, fallback, chosen, plane_fail, lp4
    int32_t *status;         // [n]
    // candidate tables (SoA [3][N]) and phi numerators
    const double *unit256, *unit128, *phi256, *phi128;
    // counters sharded 256 ways, ONE 128-BYTE LINE PER SHARD: atomics serialise per cache line at the L2
    // (64 adjacent int counters in 2 lines cost 470 us per 100k-agent step; one line each: a few us)
    int32_t *done_count;     // [256*32] count of agents not yet done after the step
    unsigned long long *agent_steps;   // [256*16] running count of agents that entered the policy (mampenv.py:35-40)
    // trajectory log [hist_cap][n], row = env step since sca_history_enable (null: off)
    HistRow *hist;
    int hist_cap, hist_row;
    int n, m, shard_begin, shard_count;
    // Whose agents are this rank's: ids [shard_begin, shard_begin + shard_count) -- or, with the cell-owner partition of
    // SCA_NBR_GRID (sca_partition.hip.h), the first shard_count entries of `present` (own == present); the entries behind them,
    // up to n_present, are the halo copies.  Null: the contiguous range, and every agent is present.
    const int32_t *own;
    const int32_t *present;
    int n_present;
    const int32_t *count_dev;   // partition mode: [0] owned, [1] halo ON THE DEVICE -- shard_count / n_present are then upper bounds the host
                                // sizes its launches with (it learns the exact counts a step or two late and never waits for them)
    int lp_kernel;           // 1: the ORCA3D-LP agents past their bootstrap step are solved by k_lp (one lane per agent)
    // k_solve in two launches (k_solve_sweep / k_solve_pick4, see solve_fast): what the first leaves for the second
    double *sw_slot;         // [n][SLOTF][K_MAX] cones / planes of the agent's neighbours, component-major (neighbour j's component q at
                             // [q][j]: the 16 lanes of a store / load touch 128 consecutive bytes)
    uint16_t *sw_surv;       // [n][512] generation indices of the table candidates outside every cone / inside every half-space
    int32_t *sw_n;           // [n] how many
    // SCA_NBR_AUTO: agents whose list the grid cannot give exactly (more than max_neighbors objects in range, or two objects of one
    // kind at the same rounded distance: the reference's order there is the kd-tree's visit order) -- queried in the kd-tree as well
    int32_t *kdq_list;       // [n]
    int32_t *kdq_count;      // [1] how many the grid query listed; more than kdq_cap: "too many for a list" -- the list is then incomplete and
    int kdq_cap;             //     the kd query of EVERY agent of the shard runs instead (k_neighbors_kd_auto)
    const AgentPar *ap;      // [n] per-agent solver attributes (null: the context's Params for everybody) -- agent_params()
    unsigned long long *kdq_stats;   // [4] AUTO passes, agents listed over them (sum, max), passes in which somebody was listed (sca_auto_stats)
    unsigned *kdq_busy;      // [1] bit 0: the grid query of this pass listed somebody and the kd query has not answered yet (the pass's stream waits for 0)
    unsigned *auto_sync;     // the launch-free form of the kd query (KdTail): [0] k_kd_block's ticket, [1] the last pass whose tree is complete, [2] the
    unsigned auto_pass_seq;  // grid query's ticket; null: this pass's kd query is a launch of its own
    int *auto_err;           // the kd build's error word (KD_ERR_SPIN: the wait for the tree gave up)
    double *trk_nbr0;        // [n] tracker in the pass: distSq of agent.neighbors[0] of THIS pass for the tracker of the next one (the
                             // epilogue saves it, so that the next pass's neighbour query may overwrite the lists while the tracker runs)
#ifdef SCA_TIMELINE
    unsigned long long *tl;  // [TL_KERNELS][TL_RING][2] first start / last end of a kernel's wavefronts, 100-MHz wall clock (sca_debug_timeline)
    int tl_step;             // the step the launch belongs to (ring index)
#endif
};
// ---- device-side timeline (DEBUG builds: SCA_BUILD_DEFS=-DSCA_TIMELINE, tools/device_timeline.py) -----------------------------------
// rocprofv3's kernel trace serialises dispatches: a chain of 5-us kernels runs ~2x slower traced than untraced (c3: 161 against 92 us per
// step, tools/timeline.py), so the un-perturbed timeline of the short steps is taken on the device instead: lane 0 of every wavefront of
// every pass kernel's first workgroup stamps the wall clock when it starts, a sample of its workgroups when they end (atomicMin / atomicMax
// into the kernel's slot of the step's ring entry).  The product build contains none of this.
enum TlKernel { TL_KD_GATHER, TL_KD_TOP, TL_KD_LEVELS, TL_KD_BLOCK, TL_GRID_COUNT, TL_GRID_FILL, TL_NBR_GRID, TL_NBR_KD, TL_NBR_KD_AUTO, TL_SOLVE, TL_SOLVE_SWEEP,
                TL_SOLVE_PICK, TL_LP, TL_FALLBACK, TL_ACTION, TL_COLLIDE, TL_GOAL_FLAGS, TL_TRACK, TL_REPLAN, TL_KERNELS };
constexpr int TL_RING = 64;
#ifdef SCA_TIMELINE
// (first form: every wavefront stamped both ends and the slot pointer lived in two VGPRs across the kernel -- 8192 same-address atomics and
// two spilled registers made k_solve 94 us instead of 12.  Now: the START is workgroup 0's first wavefront (dispatch is in order: it is the
// first to run), the END the maximum over the first threads of a SAMPLE of workgroups -- all of a launch of up to 64, else every sixteenth, the first and the last eight -- and
// the slot is recomputed from the kernel arguments, which live in SGPRs.)
struct TlScope {
    const DeviceView &d;
    const int kid;
    __device__ __forceinline__ unsigned long long *slot() const { return d.tl + 2 * ((size_t)kid * TL_RING + (size_t)(d.tl_step & (TL_RING - 1))); }
    __device__ __forceinline__ TlScope(const DeviceView &d_, int kid_) : d(d_), kid(kid_) {
        if (d.tl && blockIdx.x == 0 && threadIdx.x == 0) atomicMin(slot(), (unsigned long long)wall_clock64());
    }
    __device__ __forceinline__ ~TlScope() {
        // (denser sampling -- every wavefront of launches of <= 512 workgroups -- was tried for the kernels whose wavefronts end at very different
        // times (a wavefront per search): c3 AUTO then ran 0.108 instead of 0.088 ms per step; with this sample the burst runs at the product's pace)
        if (d.tl && threadIdx.x == 0 && (gridDim.x <= 64u || (blockIdx.x & 15u) == 15u || blockIdx.x + 8u >= gridDim.x || blockIdx.x < 8u))
            atomicMax(slot() + 1, (unsigned long long)wall_clock64());
    }
};
#define SCA_TL(d, kid) TlScope tl_scope_((d), (kid))
#else
#define SCA_TL(d, kid) do { } while (0)
#endif

// the solver attributes of one agent: the context's, or the agent's own where the swarm is heterogeneous (sca_set_agent_params)
__device__ __forceinline__ Params agent_params(const DeviceView &d, const Params &P, int agent) {
    Params Q = P;
    if (d.ap) {
        const AgentPar a = d.ap[agent];
        Q.neighbor_dist = a.neighbor_dist; Q.time_step = a.time_step; Q.time_horizon = a.time_horizon; Q.max_speed = a.max_speed;
        Q.cos_heading_thr = a.cos_heading_thr; Q.dt_nominal = a.dt_nominal; Q.max_neighbors = a.max_neighbors; Q.range_sq = a.range_sq;
    }
    return Q;
}

// the i-th agent of this rank (i < shard_count) / the i-th agent whose record this rank holds (i < present_count(d))
__device__ __forceinline__ int shard_agent(const DeviceView &d, int i) { return d.own ? d.own[i] : d.shard_begin + i; }
__device__ __forceinline__ bool shard_owns(const DeviceView &d, int agent) { return agent >= d.shard_begin && agent < d.shard_begin + d.shard_count; }
__device__ __forceinline__ int shard_size(const DeviceView &d) { return d.count_dev ? d.count_dev[0] : d.shard_count; }
__device__ __forceinline__ int present_count(const DeviceView &d) { return d.count_dev ? d.count_dev[0] + d.count_dev[1] : (d.present ? d.n_present : d.n); }
__device__ __forceinline__ int present_agent(const DeviceView &d, int i) { return d.present ? d.present[i] : i; }

// ------------------------------------------------------------------------------------------------
// wave-level helpers (64 lanes)
//
// Full-wave reductions through the DPP cross-lane path of the VALU (no LDS round trip, unlike ds_bpermute behind
// __shfl_xor): two quad permutes, two row rotations, then row_bcast:15 / row_bcast:31 fold the four 16-lane rows;
// lane 63 holds the result, which is returned wave-uniform.  All 64 lanes must be active at the call.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov(int v) {
    // with every row enabled and a pattern that gives every lane a source the old value is never used: the plain DPP move
    // needs no tied copy of the register first (one VALU instruction instead of two)
    if (ROW_MASK == 0xf && (CTRL < 0x100 || (CTRL >= 0x121 && CTRL <= 0x12f) || (CTRL >= 0x150 && CTRL <= 0x15f)))
        return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_mov64(unsigned long long v) {
    const int lo = dpp_mov<CTRL, ROW_MASK>((int)(unsigned)v), hi = dpp_mov<CTRL, ROW_MASK>((int)(unsigned)(v >> 32));
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
// inside a 16-lane row: broadcast of lane K (row_newbcast, gfx90a+), shifts towards lower / higher lanes
template <int K> __device__ __forceinline__ int row_bcast_i(int v) { return __builtin_amdgcn_mov_dpp(v, 0x150 + K, 0xf, 0xf, true); }
template <int K> __device__ __forceinline__ double row_bcast_d(double x) {
    const unsigned long long v = (unsigned long long)__double_as_longlong(x);
    const int lo = row_bcast_i<K>((int)(unsigned)v), hi = row_bcast_i<K>((int)(unsigned)(v >> 32));
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
// lane i reads lane i + K of its row
template <int K> __device__ __forceinline__ double row_shl_d(double x) {
    const unsigned long long v = (unsigned long long)__double_as_longlong(x);
    // lanes without a source read 0 (bound_ctrl): the callers only use lanes that have one
    const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)v, 0x100 + K, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)(v >> 32), 0x100 + K, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
// lane i reads lane i - 1 of its row (lane 0 reads 0)
__device__ __forceinline__ int row_shr1_i(int v) { return __builtin_amdgcn_mov_dpp(v, 0x111, 0xf, 0xf, true); }
__device__ __forceinline__ double row_shr1_d(double x) {
    const unsigned long long v = (unsigned long long)__double_as_longlong(x);
    const int lo = row_shr1_i((int)(unsigned)v), hi = row_shr1_i((int)(unsigned)(v >> 32));
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
#define SCA_DPP_REDUCE(T, MOV, OP)                                          \
    v = OP(v, (T)MOV<0xb1, 0xf>(v));  /* quad_perm [1,0,3,2] */            \
    v = OP(v, (T)MOV<0x4e, 0xf>(v));  /* quad_perm [2,3,0,1] */            \
    v = OP(v, (T)MOV<0x124, 0xf>(v)); /* row_ror:4 */                      \
    v = OP(v, (T)MOV<0x128, 0xf>(v)); /* row_ror:8 */                      \
    v = OP(v, (T)MOV<0x142, 0xa>(v)); /* row_bcast:15 into rows 1, 3 */    \
    v = OP(v, (T)MOV<0x143, 0xc>(v)); /* row_bcast:31 into rows 2, 3 */
__device__ __forceinline__ unsigned umin32(unsigned a, unsigned b) { return b < a ? b : a; }
__device__ __forceinline__ unsigned long long umin64(unsigned long long a, unsigned long long b) { return b < a ? b : a; }
__device__ __forceinline__ unsigned long long dmin_bits(unsigned long long a, unsigned long long b) {
    return __longlong_as_double((long long)b) < __longlong_as_double((long long)a) ? b : a;
}
__device__ __forceinline__ unsigned long long dmax_bits(unsigned long long a, unsigned long long b) {
    return __longlong_as_double((long long)b) > __longlong_as_double((long long)a) ? b : a;
}
__device__ __forceinline__ int iadd32(int a, int b) { return a + b; }
__device__ __forceinline__ unsigned wave_min_u32(unsigned x) {
    int v = (int)x;
    SCA_DPP_REDUCE(int, dpp_mov, [](int a, int b) { return (int)umin32((unsigned)a, (unsigned)b); })
    return (unsigned)__builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ double readlane_f64(double x, int lane_uniform) {
    const unsigned long long v = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane_uniform);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane_uniform);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ unsigned long long bcast63(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    SCA_DPP_REDUCE(unsigned long long, dpp_mov64, umin64)
    return bcast63(v);
}
__device__ __forceinline__ double wave_min_f64(double x) {
    unsigned long long v = (unsigned long long)__double_as_longlong(x);
    SCA_DPP_REDUCE(unsigned long long, dpp_mov64, dmin_bits)
    return __longlong_as_double((long long)bcast63(v));
}
__device__ __forceinline__ double wave_max_f64(double x) {
    unsigned long long v = (unsigned long long)__double_as_longlong(x);
    SCA_DPP_REDUCE(unsigned long long, dpp_mov64, dmax_bits)
    return __longlong_as_double((long long)bcast63(v));
}
// sums must not count a lane twice: the row_bcast steps add into rows that still hold only their own partial sums
// (rows 1, 3, then 2, 3), and the rotations inside a row make every lane of the row hold the row total first
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += dpp_mov<0xb1, 0xf>(v);
    v += dpp_mov<0x4e, 0xf>(v);
    v += dpp_mov<0x124, 0xf>(v);
    v += dpp_mov<0x128, 0xf>(v);                       // every lane: total of its 16-lane row
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return r0 + r1 + r2 + r3;
}
struct Key3 { double a, b; int idx; };
__device__ __forceinline__ bool key_less(const Key3 &x, const Key3 &y) {
    if (x.a < y.a) return true;
    if (x.a > y.a) return false;
    if (x.b < y.b) return true;
    if (x.b > y.b) return false;
    return x.idx < y.idx;
}
__device__ __forceinline__ Key3 key_invalid() { Key3 k; k.a = INFINITY; k.b = INFINITY; k.idx = INT_MAX; return k; }
__device__ __forceinline__ Key3 wave_argmin(Key3 k) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Key3 o;
        o.a = __shfl_xor(k.a, off);
        o.b = __shfl_xor(k.b, off);
        o.idx = __shfl_xor(k.idx, off);
        if (key_less(o, k)) k = o;
    }
    return k;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { double o = __shfl_xor(v, off); if (o < v) v = o; }
    return v;
}

// ------------------------------------------------------------------------------------------------
// K1: kd-tree neighbour query.  ONE WAVEFRONT PER AGENT: the traversal state (stack, current node) is
// wave-uniform and lives in SGPRs / LDS, node records come through the scalar cache, a leaf's <= 10 members
// are fetched and measured by 10 lanes at once, and the bounded sorted neighbour list of agent.py:87-99
// is kept one entry per lane (lanes 0..15) so that an insertion is one ballot + one lane shift.
constexpr int KD_STACK = 64;
constexpr int K1_WAVES = 4;
constexpr int NEAR_MAX = 8;

struct WaveList {            // entry k of the sorted list lives in lane k
    double dsq;
    int id;
    int cnt;                 // wave-uniform
};
// `neighbors.pop(); append; sort(key=distSq)` (agent.py:87-90): stable, so the newcomer lands after equal keys
__device__ __forceinline__ void wave_insert(WaveList &L, int lane, int maxn, int id, double dsq) {
    if (L.cnt == maxn) L.cnt--;
    const int pos = __popcll(__ballot(lane < L.cnt && L.dsq <= dsq));
    const double up_d = __shfl_up(L.dsq, 1);
    const int up_i = __shfl_up(L.id, 1);
    if (lane > pos && lane <= L.cnt) { L.dsq = up_d; L.id = up_i; }
    if (lane == pos) { L.dsq = dsq; L.id = id; }
    L.cnt++;
}

// Depth-first traversal of kdTree.py:127-156 (nearer child first, ties go right, constant rangeSq) with a
// wave-uniform explicit stack.  leaf(begin, end) is called for every visited leaf in visit order.
__device__ __forceinline__ double box_dist_sq(const KdWide &w, int side, V3 p) {   // kdTree.py:132-145
    double t, s;
    t = fmax(0.0, w.bx[kdw_idx(side, 0, 0)] - p.x); s = t * t;
    t = fmax(0.0, p.x - w.bx[kdw_idx(side, 1, 0)]); s = s + t * t;
    t = fmax(0.0, w.bx[kdw_idx(side, 0, 1)] - p.y); s = s + t * t;
    t = fmax(0.0, p.y - w.bx[kdw_idx(side, 1, 1)]); s = s + t * t;
    t = fmax(0.0, w.bx[kdw_idx(side, 0, 2)] - p.z); s = s + t * t;
    t = fmax(0.0, p.z - w.bx[kdw_idx(side, 1, 2)]); s = s + t * t;
    return s;
}

template <class LeafFn>
__device__ __forceinline__ int kd_traverse(const KdWide *tree, V3 p, double rangeSq, int *stack, int lane, LeafFn leaf) {
    int sp = 0, st = 0;
    int node = 0;
    bool have = true;
    while (have) {
        node = __builtin_amdgcn_readfirstlane(node);
        const KdWide nd = tree[node];
        have = false;
        if (nd.end - nd.begin <= MAX_LEAF) {
            leaf(nd.begin, nd.end);
        } else {
            const double dl = box_dist_sq(nd, 0, p);
            const double dr = box_dist_sq(nd, 1, p);
            int first, second;
            double dfirst, dsecond;
            if (dl < dr) { first = nd.left; second = nd.right; dfirst = dl; dsecond = dr; }
            else { first = nd.right; second = nd.left; dfirst = dr; dsecond = dl; }
            if (dfirst < rangeSq) {
                if (dsecond < rangeSq) {
                    if (sp < KD_STACK) { if (lane == 0) stack[sp] = second; sp++; }
                    else st |= ST_KD_STACK;
                }
                node = first;
                have = true;
            }
        }
        if (!have && sp > 0) {
            sp--;
            __builtin_amdgcn_wave_barrier();
            node = stack[sp];
            have = true;
        }
    }
    return st;
}

__device__ __forceinline__ int lo32(double v) { return (int)(unsigned)((unsigned long long)__double_as_longlong(v) & 0xffffffffull); }
__device__ __forceinline__ int hi32(double v) { return (int)(unsigned)((unsigned long long)__double_as_longlong(v) >> 32); }

// The same traversal for the one-agent-per-wavefront neighbour kernel, which is pure node-fetch latency at small N:
// the 16 lanes of row 0 hold the current node's 128-byte record (one double each, as in the packed kernel), and when both
// children will be visited ONE load instruction brings both records: row 0 takes the child visited first, row 1 the other
// one, which goes onto a stack of RECORDS in LDS -- popping it later costs an LDS read instead of another trip to the L2.
constexpr int KD_RSTACK = 48;
template <class LeafFn>
__device__ __forceinline__ int kd_traverse_rec(const KdWide *tree, V3 p, double rangeSq, double (*rstack)[16], int lane, LeafFn leaf) {
    const double *wd = (const double *)tree;
    const int gl = lane & 15, row = lane >> 4;
    const int tk = gl >= 2 ? (gl - 2) >> 2 : 0;                       // which box term this lane squares (see k_neighbors_kd4)
    const bool t_is_mx = gl >= 2 && (((gl - 2) >> 1) & 1);
    const bool t_live = gl >= 2 && gl < 14;
    const double pk = tk == 0 ? p.x : (tk == 1 ? p.y : p.z);
    int sp = 0, st = 0;
    double w = lane < 16 ? wd[lane] : 0.0;                            // the root's record
    bool have = true;
    while (have) {
        const double h0 = readlane_f64(w, 0), h1 = readlane_f64(w, 1);
        const int nb = lo32(h0), ne = hi32(h0), nl = lo32(h1), nr = hi32(h1);
        bool descend = false, push = false;
        int first = 0, second = 0;
        if (ne - nb <= MAX_LEAF) {
            leaf(nb, ne);
        } else {                                                      // kdTree.py:132-156
            double t = t_is_mx ? pk - w : w - pk;
            t = fmax(0.0, t);
            const double sq = t_live ? t * t : 0.0;
            double ssum = sq;                                         // lane 2: left child's terms in order, lane 3: right child's
            ssum = ssum + row_shl_d<2>(sq);
            ssum = ssum + row_shl_d<4>(sq);
            ssum = ssum + row_shl_d<6>(sq);
            ssum = ssum + row_shl_d<8>(sq);
            ssum = ssum + row_shl_d<10>(sq);
            const double dl = readlane_f64(ssum, 2), dr = readlane_f64(ssum, 3);
            double dfirst, dsecond;
            if (dl < dr) { first = nl; second = nr; dfirst = dl; dsecond = dr; }
            else { first = nr; second = nl; dfirst = dr; dsecond = dl; }
            if (dfirst < rangeSq) { descend = true; push = dsecond < rangeSq; }
        }
        if (descend) {
            const bool ld = row == 0 || (row == 1 && push);
            const double nw = ld ? wd[(size_t)(row == 0 ? first : second) * 16 + gl] : 0.0;
            if (push) {
                if (sp < KD_RSTACK) { if (row == 1) rstack[sp][gl] = nw; sp++; }
                else st |= ST_KD_STACK;
            }
            w = nw;
        } else if (sp > 0) {
            sp--;
            __builtin_amdgcn_wave_barrier();
            w = lane < 16 ? rstack[sp][lane] : 0.0;
        } else have = false;
    }
    return st;
}

// agent_reach / obs_reach: see k_collide_finish.  Every object that can touch this agent after the move is visited
// here anyway (it is within neighborDist), so the few that are close enough are written down for K4.
// HAS_OBS (round 6): the obstacle phase squares a surface distance with libm's pow (agent.py:106) -- a call whose register window (62 ->
// 80 VGPRs in k_neighbors_kd4) a scene WITHOUT obstacles should not pay for: the host launches the <false> form there (no obstacle code).
template <bool HAS_OBS = true>
__device__ __forceinline__ void neighbors_one(const DeviceView &d, const Params &P, double agent_reach, double obs_reach,
                                              double max_radius, double (*rstack)[16], int agent, int lane) {
    const PubRec me = d.rec[agent];
    int st = 0;
    bool skip = (me.flags & (FLAG_AT_GOAL | FLAG_COLLISION | FLAG_TIMEOUT)) != 0;   // mampenv.py:35
    const int pol = d.policy[agent];
    const bool orca = (pol == POL_ORCA || pol == POL_ORCA_LP);
    F3 vA; vA.x = me.vx; vA.y = me.vy; vA.z = me.vz;
    // SCA / RVO / S-RVO skip computeNeighbors on the bootstrap step (scaPolicy.py:34); ORCA does not (orca3dPolicy.py:51)
    if (!orca && l3norm_f32zero(vA, false) <= 1e-5) skip = true;
    if (skip) {
        if (lane < K_MAX) { d.nbr_id[agent * K_MAX + lane] = -1; d.nbr_dsq[agent * K_MAX + lane] = 0.0; }
        if (lane == 0) { d.coll_new[agent] = 0; d.nbr_valid[agent] = 0; d.nbr_n[agent] = 0; d.status[agent] = 0; d.near_n[agent] = -1; }
        return;
    }
    int near_cnt = 0;                    // wave-uniform
    int *near_out = d.near_id + (size_t)agent * NEAR_MAX;
    const double reach_a = me.radius + agent_reach, reach_o = me.radius + obs_reach;
    const V3 pA = v3(me.px, me.py, me.pz);
    const double rangeSq = d.ap ? d.ap[agent].range_sq : P.range_sq;              // scaPolicy.py:112 (the agent's own where the swarm is heterogeneous)
    const int maxn = d.ap ? d.ap[agent].max_neighbors : P.max_neighbors;
    WaveList L; L.dsq = 0.0; L.id = -1; L.cnt = 0;
    bool coll = false;
    // obstacles first (scaPolicy.py:114-116), agent.py:101-124
    if (HAS_OBS && d.m > 0) {
        st |= kd_traverse_rec(d.owide, pA, rangeSq, rstack, lane, [&](int begin, int end) {
            const bool valid = lane < end - begin;
            int o = 0; double distSq = 0.0; bool c = false, r = false, nr = false;
            if (valid) {
                o = d.operm[begin + lane];
                const ObsRec ob = d.obs_sorted[begin + lane];
                const V3 pO = v3(ob.px, ob.py, ob.pz);
                const double distSq1 = l3normsq(pA, pO);
                const double t = l3norm(pA, pO) - ob.radius;
                distSq = m_pow2(t);                                  // (l3norm(pA, pO) - r) ** 2 = libm's pow (agent.py:106): the list's distSq is the reference's bits
                const double rs = me.radius + ob.radius;
                r = distSq < rangeSq;
                c = r && distSq1 < rs * rs;
                const V3 dd = pA - pO;
                nr = (dd.x * dd.x + dd.y * dd.y + dd.z * dd.z) < reach_o * reach_o;
            }
            {
                const unsigned long long nm = __ballot(nr);
                if (nr) { const int at = near_cnt + __popcll(nm & ((1ull << lane) - 1ull)); if (at < NEAR_MAX) near_out[at] = o | NBR_OBSTACLE_BIT; }
                near_cnt += __popcll(nm);
            }
            unsigned long long todo = __ballot(r);
            while (todo) {
                const int b = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const bool cb = __shfl((int)c, b) != 0;
                const double db = __shfl(distSq, b);
                const int ib = __shfl(o, b) | NBR_OBSTACLE_BIT;
                if (cb) { if (!coll) { coll = true; L.cnt = 0; } wave_insert(L, lane, maxn, ib, db); }
                else if (!coll) wave_insert(L, lane, maxn, ib, db);
            }
        });
    }
    // other agents, agent.py:79-99
    // leaf members are contiguous in position order: ids and coordinates come in one coalesced round trip; the other
    // agent's radius is only fetched when the pair is close enough for the collision test to matter
    const double rmax2 = (me.radius + max_radius) * (me.radius + max_radius);
    st |= kd_traverse_rec(d.awide, pA, rangeSq, rstack, lane, [&](int begin, int end) {
        const bool valid = lane < end - begin;
        int o = -1; double distSq = 0.0; bool c = false, r = false, nr = false;
        if (valid) {
            o = d.aperm[begin + lane];
            if (o != agent) {
                distSq = l3normsq(pA, v3(d.kx[begin + lane], d.ky[begin + lane], d.kz[begin + lane]));
                r = distSq < rangeSq;
                if (r && distSq < rmax2) { const double rs = me.radius + d.rec[o].radius; c = distSq < rs * rs; }
                nr = distSq < reach_a * reach_a;
            }
        }
        {
            const unsigned long long nm = __ballot(nr);
            if (nr) { const int at = near_cnt + __popcll(nm & ((1ull << lane) - 1ull)); if (at < NEAR_MAX) near_out[at] = o; }
            near_cnt += __popcll(nm);
        }
        unsigned long long todo = __ballot(r);
        while (todo) {
            const int b = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const bool cb = __shfl((int)c, b) != 0;
            const double db = __shfl(distSq, b);
            const int ib = __shfl(o, b);
            if (cb) { if (!coll) { coll = true; L.cnt = 0; } wave_insert(L, lane, maxn, ib, db); }
            else if (!coll) wave_insert(L, lane, maxn, ib, db);
        }
    });
    if (lane < K_MAX) {
        d.nbr_id[agent * K_MAX + lane] = (lane < L.cnt) ? L.id : -1;
        d.nbr_dsq[agent * K_MAX + lane] = (lane < L.cnt) ? L.dsq : 0.0;
    }
    if (lane == 0) {
        d.nbr_n[agent] = L.cnt;
        d.nbr_valid[agent] = 1;
        d.coll_new[agent] = coll ? 1u : 0u;
        d.status[agent] = st;
        // candidates are only complete if the whole reach lies inside the visited range
        const bool complete = near_cnt <= NEAR_MAX && reach_a * reach_a <= rangeSq && reach_o * reach_o <= rangeSq;
        d.near_n[agent] = complete ? near_cnt : -1;
    }
}

template <bool HAS_OBS>
__global__ __launch_bounds__(K1_WAVES * 64) void k_neighbors_kd(DeviceView d, Params P, double agent_reach, double obs_reach,
                                                                double max_radius) {
    SCA_TL(d, TL_NBR_KD);
    __shared__ double rstacks[K1_WAVES][KD_RSTACK][16];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int agent = d.shard_begin + blockIdx.x * K1_WAVES + wid;
    if (agent < d.shard_begin + d.shard_count) neighbors_one<HAS_OBS>(d, P, agent_reach, obs_reach, max_radius, rstacks[wid], agent, lane);
}

// SCA_NBR_AUTO: the kd query behind the build, ONE launch on the build's stream: nobody listed by the grid query (the usual case) --
// nothing to do; a list of up to kdq_cap agents -- one wavefront each, a fixed grid striding over the list; more ("too many for a list":
// the list is incomplete) -- every agent of the shard the same way, i.e. a whole kd pass's query.  Overwrites everything the grid query
// left for the agents it queries.
// The hand-shake: one word in device memory, bit 0 = "the grid query of this pass listed somebody and the kd query has not answered
// yet" -- set by the grid query's listing wavefronts, cleared by the last workgroup here.  The stream that goes on to the solve waits
// for the bit to be clear with hipStreamWaitValue32: a wait that costs nothing when nobody was listed, which is what keeps the kd
// build (beside, on its own stream) off the pass's critical path in the common case.  (Until late in round 4 a word counting passes,
// and a one-lane launch behind the grid query to advance it when nobody was listed: 4 us + its gap on every pass's path.)
constexpr int KDQ_BLOCKS = 1024, KDQ_BLOCKS_FEW = 64;   // (the few: while the counts that came back say a wavefront each is enough -- an empty launch of 64 workgroups is half as long)
__global__ __launch_bounds__(K1_WAVES * 64) void k_neighbors_kd_auto(DeviceView d, Params P, double agent_reach, double obs_reach,
                                                                     double max_radius, int *ticket) {
    SCA_TL(d, TL_NBR_KD_AUTO);
    __shared__ double rstacks[K1_WAVES][KD_RSTACK][16];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = *d.kdq_count;
    if (d.kdq_stats && blockIdx.x == 0 && threadIdx.x == 0) {           // statistics: passes, listed agents (sum, max), passes with somebody listed
        d.kdq_stats[0] += 1; d.kdq_stats[1] += (unsigned long long)n;
        if ((unsigned long long)n > d.kdq_stats[2]) d.kdq_stats[2] = (unsigned long long)n;
        if (n > 0) d.kdq_stats[3] += 1;
    }
    if (n == 0) return;
    if (n <= d.kdq_cap) {
        for (int i = (int)blockIdx.x * K1_WAVES + wid; i < n; i += (int)gridDim.x * K1_WAVES)
            neighbors_one(d, P, agent_reach, obs_reach, max_radius, rstacks[wid], d.kdq_list[i], lane);
    } else {
        for (int i = (int)blockIdx.x * K1_WAVES + wid; i < d.shard_count; i += (int)gridDim.x * K1_WAVES)
            neighbors_one(d, P, agent_reach, obs_reach, max_radius, rstacks[wid], d.shard_begin + i, lane);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(ticket, 1) == (int)gridDim.x - 1) { *ticket = 0; __threadfence(); atomicAnd(d.kdq_busy, ~1u); }
    }
}

// ------------------------------------------------------------------------------------------------
// K1, packed form: FOUR AGENTS PER WAVEFRONT, 16 lanes each.  The box arithmetic of an inner node is the same for all
// 64 lanes in the one-agent-per-wave kernel (36 fp64 operations per node, per wave); here the 16 lanes of a group load
// the 16 doubles of the node record in one 128-byte read, 12 of them square one box term each, and the ordered sums
// (kdTree.py:132-145 adds the terms in a fixed order) are gathered on two lanes -- for four agents at once.
// Sixteen lanes are also exactly what the rest needs: <= 10 leaf members, <= 16 list entries (agent.py:32).
constexpr int K1P_WAVES = 4;
constexpr int K1P_G = 16;
constexpr int K1P_APW = 4;


// In a tracked pass K1 runs beside the re-plan kernel, whose wavefronts are older and therefore served first by the SIMD's
// arbiter: 293 us there against 90 alone, and the neighbour branch (K0 -> K1 -> k_solve_sweep) ended last.  With issue priority
// it takes what it needs (the re-plan kernel's longest wavefronts mostly sit alone on their SIMDs): c4 0.740 -> 0.718 ms.
// k_solve_sweep gets none: with it the re-plans end 90 us later (measured: 0.775 ms).  (SCA_K1_PRIO=0 switches it off.)
#ifndef SCA_K1_PRIO
#define SCA_K1_PRIO 3
#endif
#if SCA_K1_PRIO > 0
#define SCA_K1_SETPRIO() __builtin_amdgcn_s_setprio(SCA_K1_PRIO)
#else
#define SCA_K1_SETPRIO() ((void)0)
#endif
template <bool HAS_OBS>
__global__ __launch_bounds__(K1P_WAVES * 64) void k_neighbors_kd4(DeviceView d, Params P, double agent_reach, double obs_reach,
                                                                 double max_radius) {
    SCA_TL(d, TL_NBR_KD);
    SCA_K1_SETPRIO();
    __shared__ int stacks[K1P_WAVES][K1P_APW][KD_STACK];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int g = lane >> 4, gl = lane & 15, gshift = g << 4;
    const int end = d.shard_begin + d.shard_count;
    const int agent_raw = d.shard_begin + (blockIdx.x * K1P_WAVES + wid) * K1P_APW + g;
    const bool exists = agent_raw < end;
    const int agent = exists ? agent_raw : end - 1;                 // clamp: idle groups read a valid record, write nothing
    const PubRec me = d.rec[agent];
    int st = 0;
    bool skip = !exists || (me.flags & (FLAG_AT_GOAL | FLAG_COLLISION | FLAG_TIMEOUT)) != 0;   // mampenv.py:35
    const int pol = d.policy[agent];
    const bool orca = (pol == POL_ORCA || pol == POL_ORCA_LP);
    F3 vA; vA.x = me.vx; vA.y = me.vy; vA.z = me.vz;
    if (!orca && l3norm_f32zero(vA, false) <= 1e-5) skip = true;    // scaPolicy.py:34: no computeNeighbors on the bootstrap step
    const V3 pA = v3(me.px, me.py, me.pz);
    const double rangeSq = d.ap ? d.ap[agent].range_sq : P.range_sq;         // scaPolicy.py:112: neighborDist ** 2, the agent's own where the swarm is heterogeneous
    const int maxn = d.ap ? d.ap[agent].max_neighbors : P.max_neighbors;
    const double reach_a = me.radius + agent_reach, reach_o = me.radius + obs_reach;
    const double rmax2 = (me.radius + max_radius) * (me.radius + max_radius);
    // which box term this lane squares: lanes 2..13 hold bx[0..11] of the node record (axis, min/max, side interleaved)
    const int tk = gl >= 2 ? (gl - 2) >> 2 : 0;
    const bool t_is_mx = gl >= 2 && (((gl - 2) >> 1) & 1);
    const bool t_live = gl >= 2 && gl < 14;
    const double pk = tk == 0 ? pA.x : (tk == 1 ? pA.y : pA.z);
    // sorted list: entry gl of the group's agent
    double Ld = 0.0; int Li = -1; int cnt = 0; bool coll = false; int near_cnt = 0;
    int *near_out = d.near_id + (size_t)agent * NEAR_MAX;
    int *stack = stacks[wid][g];

    // (HAS_OBS = false, launched for scenes without obstacles, keeps the loop's shape -- the obstacle phase stays in the code, with x * x,
    // and is never taken: without it the compiler turned 1.2 M scalar instructions per launch into 6.7 M vector ones at c4, PMC-measured)
    for (int phase = 0; phase < 2; phase++) {                       // obstacles first (scaPolicy.py:114-116)
        const bool ob = phase == 0;
        if (ob && d.m <= 0) continue;
        const double *wd = (const double *)(ob ? d.owide : d.awide);
        int node = 0, sp = 0;
        bool have = !skip;
        while (__any(have)) {
            const double w = have ? wd[(size_t)node * 16 + gl] : 0.0;
            const double h0 = row_bcast_d<0>(w), h1 = row_bcast_d<1>(w);
            const int nb = lo32(h0), ne = hi32(h0), nl = lo32(h1), nr_ = hi32(h1);
            const bool leaf = have && (ne - nb <= MAX_LEAF);
            const bool inner = have && !leaf;
            bool descend = false; int next = 0;
            if (__any(inner)) {                                      // kdTree.py:132-156
                double t = t_is_mx ? pk - w : w - pk;
                t = fmax(0.0, t);
                const double sq = t_live ? t * t : 0.0;
                // lane 2 adds the left child's terms mn0 mx0 mn1 mx1 mn2 mx2 in that order, lane 3 the right child's
                double ssum = sq;
                ssum = ssum + row_shl_d<2>(sq);
                ssum = ssum + row_shl_d<4>(sq);
                ssum = ssum + row_shl_d<6>(sq);
                ssum = ssum + row_shl_d<8>(sq);
                ssum = ssum + row_shl_d<10>(sq);
                const double dl = row_bcast_d<2>(ssum), dr = row_bcast_d<3>(ssum);
                int first, second; double dfirst, dsecond;
                if (dl < dr) { first = nl; second = nr_; dfirst = dl; dsecond = dr; }
                else { first = nr_; second = nl; dfirst = dr; dsecond = dl; }
                if (inner && dfirst < rangeSq) {
                    if (dsecond < rangeSq) {
                        if (sp < KD_STACK) { if (gl == 0) stack[sp] = second; sp++; }
                        else st |= ST_KD_STACK;
                    }
                    next = first; descend = true;
                }
            }
            if (__any(leaf)) {
                const bool valid = leaf && gl < ne - nb;
                int o = -1; double dsq = 0.0; bool c = false, r = false, nr = false;
                if (valid) {
                    if (ob) {                                        // agent.py:101-124
                        o = d.operm[nb + gl];
                        const ObsRec orec = d.obs_sorted[nb + gl];
                        const V3 pO = v3(orec.px, orec.py, orec.pz);
                        const double distSq1 = l3normsq(pA, pO);
                        const double tt = l3norm(pA, pO) - orec.radius;
                        dsq = HAS_OBS ? m_pow2(tt) : tt * tt;    // (... ) ** 2 = libm's pow (agent.py:106); <false>: never reached
                        const double rs = me.radius + orec.radius;
                        r = dsq < rangeSq;
                        c = r && distSq1 < rs * rs;
                        const V3 dd = pA - pO;
                        nr = (dd.x * dd.x + dd.y * dd.y + dd.z * dd.z) < reach_o * reach_o;
                        o |= NBR_OBSTACLE_BIT;
                    } else {                                         // agent.py:79-99
                        o = d.aperm[nb + gl];
                        if (o != agent) {
                            dsq = l3normsq(pA, v3(d.kx[nb + gl], d.ky[nb + gl], d.kz[nb + gl]));
                            r = dsq < rangeSq;
                            if (r && dsq < rmax2) { const double rs = me.radius + d.rec[o].radius; c = dsq < rs * rs; }
                            nr = dsq < reach_a * reach_a;
                        }
                    }
                }
                {
                    const unsigned nm = (unsigned)((__ballot(nr) >> gshift) & 0xffffull);
                    if (nr) { const int at = near_cnt + __popc(nm & ((1u << gl) - 1u)); if (at < NEAR_MAX) near_out[at] = o; }
                    near_cnt += __popc(nm);
                }
                const unsigned todo = (unsigned)((__ballot(r) >> gshift) & 0xffffull);
                // members in range, in leaf order: member k of every group's leaf is broadcast inside the group's row
                auto member = [&](bool act, bool cb, double db, int ib) {
                    if (act && cb && !coll) { coll = true; cnt = 0; }            // agent.py:83-85
                    const bool ins = act && (cb || !coll);
                    int ncnt = cnt;
                    if (ins && ncnt == maxn) ncnt--;                             // neighbors.pop()
                    const unsigned bm = (unsigned)((__ballot(ins && gl < ncnt && Ld <= db) >> gshift) & 0xffffull);
                    const int pos = __popc(bm);
                    const double up_d = row_shr1_d(Ld);
                    const int up_i = row_shr1_i(Li);
                    if (ins) {
                        if (gl > pos && gl <= ncnt) { Ld = up_d; Li = up_i; }
                        if (gl == pos) { Ld = db; Li = ib; }
                        cnt = ncnt + 1;
                    }
                };
                const int ci = c ? 1 : 0;
#define SCA_K1_MEMBER(KK)                                                                                   \
                if (__any((todo >> KK) & 1u)) member(((todo >> KK) & 1u) != 0, row_bcast_i<KK>(ci) != 0, row_bcast_d<KK>(dsq), row_bcast_i<KK>(o));
                SCA_K1_MEMBER(0) SCA_K1_MEMBER(1) SCA_K1_MEMBER(2) SCA_K1_MEMBER(3) SCA_K1_MEMBER(4)
                SCA_K1_MEMBER(5) SCA_K1_MEMBER(6) SCA_K1_MEMBER(7) SCA_K1_MEMBER(8) SCA_K1_MEMBER(9)
#undef SCA_K1_MEMBER
                static_assert(MAX_LEAF == 10, "the member sequence above is written for leaves of <= 10");
            }
            if (have) {
                if (descend) node = next;
                else if (sp > 0) { sp--; node = stack[sp]; }
                else have = false;
            }
        }
    }
    if (!exists) return;
    if (skip) {
        d.nbr_id[agent * K_MAX + gl] = -1; d.nbr_dsq[agent * K_MAX + gl] = 0.0;
        if (gl == 0) { d.coll_new[agent] = 0; d.nbr_valid[agent] = 0; d.nbr_n[agent] = 0; d.status[agent] = 0; d.near_n[agent] = -1; }
        return;
    }
    d.nbr_id[agent * K_MAX + gl] = (gl < cnt) ? Li : -1;
    d.nbr_dsq[agent * K_MAX + gl] = (gl < cnt) ? Ld : 0.0;
    if (gl == 0) {
        d.nbr_n[agent] = cnt;
        d.nbr_valid[agent] = 1;
        d.coll_new[agent] = coll ? 1u : 0u;
        d.status[agent] = st;
        const bool complete = near_cnt <= NEAR_MAX && reach_a * reach_a <= rangeSq && reach_o * reach_o <= rangeSq;
        d.near_n[agent] = complete ? near_cnt : -1;
    }
}

// ------------------------------------------------------------------------------------------------
// K2/K3: one wavefront per agent.
constexpr int SOLVE_WAVES = 4;
constexpr int SLOT = 16;                  // doubles per neighbour slot in LDS (k_solve_full: cone + R, absSq for time-to-collision)
constexpr int SLOTF = 8;                  // k_solve: apex / point (3), pAB / normal (3), g
constexpr int NR = 8;                     // candidate rounds: 8 * 64 = 512 table candidates

// Per-agent scalar prologue of find_next_action, computed ONE LANE PER AGENT (inside k_kd_gather, or k_prep when the
// tree comes from the host) instead of by a whole wavefront in k_solve: preferred velocity, bootstrap test, the second
// candidate speed, |vA|, and everything about the v_pref candidate that does not depend on the neighbours.
struct alignas(16) Prep {
    double vpref[3];
    double nvA;            // |vA| as float32 norm (util.py:11)
    double rad1;           // second element of np.arange(0.5, ps + 0.03, ps - 0.5)
    unsigned vp_key;       // round5 numerator of |v_pref - v_pref| (= 0) << 10, without the index
    unsigned bits;         // 1 first_step, 2 bad pref speed, 4 v_pref passes the posture constraint, (8 unused),
                           // 16 the straight-line v_pref sits on a rounding edge; bits 8..: get_phi
                           // numerator of v_pref
};
static_assert(sizeof(Prep) == 48, "Prep must be 48 bytes");

// the agents whose v_pref the device tracker computes in this pass (mampenv.py:35 + the policy): their prologue is written by the
// tracker's kernels right behind v_pref (track_store), everybody else's by the neighbour structure's first kernel
__device__ __forceinline__ bool tracker_owns(const DeviceView &d, int agent) {
    const int pol = d.policy[agent];
    return (pol == POL_SCA || pol == POL_RVO_DUBINS) && (d.rec[agent].flags & 7u) == 0u;
}
#ifndef SCA_PREP_LIBM
#define SCA_PREP_LIBM 2             // the prologue kernels' arctangent: 2 inline on the constant tables, 0 a call (A/B)
#endif
constexpr int PREP_LIBM = SCA_PREP_LIBM;
template <int LIBM = 0>
__device__ __forceinline__ void prep_agent(const DeviceView &d, const Params &Pctx, Prep *out, int agent) {
    const Params P = agent_params(d, Pctx, agent);
    const PubRec me = d.rec[agent];
    const int pol = d.policy[agent];
    const bool orca = (pol == POL_ORCA || pol == POL_ORCA_LP);
    const V3 pA = v3(me.px, me.py, me.pz);
    F3 vA; vA.x = me.vx; vA.y = me.vy; vA.z = me.vz;
    const double ps = d.pref_speed[agent];
    V3 vpref;
    if (d.vpref_mode[agent]) vpref = v3(d.vpref_ext[agent * 3], d.vpref_ext[agent * 3 + 1], d.vpref_ext[agent * 3 + 2]);
    else vpref = straight_v_pref(v3(d.goal[agent * 3], d.goal[agent * 3 + 1], d.goal[agent * 3 + 2]), pA, ps, orca);
    Prep r;
    r.vpref[0] = vpref.x; r.vpref[1] = vpref.y; r.vpref[2] = vpref.z;
    r.nvA = (double)normf(vA);
    unsigned bits = 0;
    if (l3norm_f32zero(vA, orca) <= 1e-5) bits |= 1u;                                // scaPolicy.py:34 / orca3dPolicy.py:53
    double rad1;
    if (!candidate_speeds(ps, rad1)) { bits |= 2u; rad1 = ps; }
    r.rad1 = rad1;
    if (posture_ok(P, vA, r.nvA, pA.z, vpref)) bits |= 4u;
    double kn;
    l3norm(vpref, vpref, &kn);
    r.vp_key = pack_key(kn, 0);
    // util.py:145 of the v_pref candidate (<= 628318): read by shunted_strategy only (SCA, S-RVO3D) -- the other policies' prologue skips
    // the atan2 (a call into the restated libm since round 6)
    if (pol == POL_SCA || pol == POL_SRVO) bits |= (unsigned)get_phi_num<LIBM>(vpref.x, vpref.y) << 8;
    r.bits = bits;
    out[agent] = r;
}

struct SolveLds {
    double slot[SOLVE_WAVES][K_MAX][SLOT];
    Plane planes[SOLVE_WAVES][K_MAX];
    Plane proj[SOLVE_WAVES][K_MAX];
};

struct CandTab {
    const double *unit;     // SoA [3][num_N]
    const double *phi;      // [num_N] phi numerators of the unit directions
    int num_N;
    double rad1;
    int vp_idx;             // generation index of the v_pref candidate (= 2 * num_N)
};
__device__ __forceinline__ V3 cand_from_idx(const CandTab &T, int idx, V3 vpref) {
    if (idx >= T.vp_idx) return vpref;
    const int n0 = (idx >= T.num_N) ? idx - T.num_N : idx;
    const double rad = (idx >= T.num_N) ? T.rad1 : 0.5;
    return v3(rad * T.unit[n0], rad * T.unit[T.num_N + n0], rad * T.unit[2 * T.num_N + n0]);
}
__device__ __forceinline__ double phi_from_idx(const CandTab &T, int idx, V3 vpref) {
    if (idx >= T.vp_idx) return get_phi_num(vpref.x, vpref.y);
    const int n0 = (idx >= T.num_N) ? idx - T.num_N : idx;
    return T.phi[n0];
}

// update_velocitie (mampenv.py:83-105) for one agent: the moved record goes to rec_new (positions of the
// current step stay readable for everybody else until the host swaps the buffers).
template <bool INLINE_LIBM = false>
__device__ __forceinline__ void integrate_agent(const DeviceView &d, const Params &P, int agent, PubRec r, const float *act) {
    const double speed = (double)act[3];
    const double a = pi_2_pi(d.heading[agent * 3 + 0] + (double)act[4]);
    const double b = pi_2_pi(d.heading[agent * 3 + 1] + (double)act[5]);
    const double g = pi_2_pi(d.heading[agent * 3 + 2] + (double)act[6]);
    const double dt = d.ap ? d.ap[agent].dt_nominal : P.dt_nominal;               // agent.dt_nominal (mampenv.py:90-92)
    double sa, ca, sb, cb;                                                        // math.sin / math.cos: the restated glibc (sca_core.h)
    if (INLINE_LIBM) { m_sincos_i(a, sa, ca); m_sincos_i(b, sb, cb); }       // (k_action: the two overlap; the 250-register fallback sweep calls)
    else { m_sincos(a, sa, ca); m_sincos(b, sb, cb); }
    const double dx = speed * cb * ca * dt;
    const double dy = speed * cb * sa * dt;
    const double dz = speed * sb * dt;
    const double len = INLINE_LIBM ? sqrt(m_pow2_i(dx) + m_pow2_i(dy) + m_pow2_i(dz)) : sqrt(m_pow2(dx) + m_pow2(dy) + m_pow2(dz));               // sqrt(dx ** 2 + dy ** 2 + dz ** 2), mampenv.py:94
    d.total_dist[agent] += len;
    r.px += dx; r.py += dy; r.pz += dz;
    r.vx = act[0]; r.vy = act[1]; r.vz = act[2];
    d.heading[agent * 3 + 0] = a; d.heading[agent * 3 + 1] = b; d.heading[agent * 3 + 2] = g;
    if (!(r.flags & FLAG_AT_GOAL)) d.step_num[agent] += 1;
    d.rec_new[agent] = r;
    if (d.hist && d.hist_row < d.hist_cap) {          // mampenv.py:105 agent.to_vector()
        HistRow h;
        h.px = r.px; h.py = r.py; h.pz = r.pz; h.a = a; h.b = b; h.g = g;
        h.vx = r.vx; h.vy = r.vy; h.vz = r.vz; h.flags = r.flags;
        d.hist[(size_t)d.hist_row * d.n + agent] = h;
    }
}

// generalized pass-on-the-right (scaPolicy.py:119-145) on a list given as per-lane slots.
// inlist/key/idx: NR table slots per lane + one extra slot (v_pref) that only lane 0 owns.
// Order of the reference's sorted list == lexicographic (key, idx).
__device__ __forceinline__ int select_from_list(bool shunted, double thr, int count, const bool inl[NR + 1], const double key[NR + 1],
                                const int idx[NR + 1], const V3 cand[NR + 1], const CandTab &T, V3 vpref, V3 vA64) {
    Key3 best = key_invalid();
#pragma unroll
    for (int r = 0; r <= NR; r++)
        if (inl[r]) { Key3 k; k.a = key[r]; k.b = 0.0; k.idx = idx[r]; if (key_less(k, best)) best = k; }
    best = wave_argmin(best);
    if (!shunted || count <= 1) return best.idx;
    const V3 c0 = cand_from_idx(T, best.idx, vpref);
    const double s0 = l3norm(c0, vA64);
    // first element (in sorted order) that breaks the prefix
    bool pass[NR + 1];
    Key3 fail = key_invalid();
#pragma unroll
    for (int r = 0; r <= NR; r++) {
        pass[r] = false;
        if (inl[r]) {
            const double s = l3norm(cand[r], vA64);
            pass[r] = fabs(s0 - s) < thr;
            if (!pass[r]) { Key3 k; k.a = key[r]; k.b = 0.0; k.idx = idx[r]; if (key_less(k, fail)) fail = k; }
        }
    }
    fail = wave_argmin(fail);
    // vA_phi_min / vA_phi_max: first extremal phi in sorted order
    Key3 kmin = key_invalid(), kmax = key_invalid();
#pragma unroll
    for (int r = 0; r <= NR; r++) {
        if (inl[r]) {
            Key3 me; me.a = key[r]; me.b = 0.0; me.idx = idx[r];
            if (key_less(me, fail)) {
                const double ph = phi_from_idx(T, idx[r], vpref);
                Key3 a; a.a = ph; a.b = key[r]; a.idx = idx[r];
                Key3 b; b.a = -ph; b.b = key[r]; b.idx = idx[r];
                if (key_less(a, kmin)) kmin = a;
                if (key_less(b, kmax)) kmax = b;
            }
        }
    }
    kmin = wave_argmin(kmin);
    kmax = wave_argmin(kmax);
    const double phi_min = kmin.a / EPS5, phi_max = (-kmax.a) / EPS5;
    if (fabs(phi_max - phi_min) <= PI) return kmin.idx;
    return kmax.idx;
}

// The hot loop: K neighbours x NROUND candidates per lane, branch-free.  Per pair (cone): 3 sub, 2x(mul+2 fma), 2 mul,
// 2 compares -- all fp64, no transcendental (the asin/acos comparison of util.py:30-41 in algebraic form).
// EARLY: leave the loop once every candidate of the wavefront is dead (worth it for the 8-round form of the complete
// sweep; in k_solve's short groups the test costs more scalar work per trip than the rare early exit saves).
template <int NROUND, bool ORCA, int NA, int NB, int SL, bool EARLY = true>
__device__ __forceinline__ unsigned sweep(const double (*slot)[SL], int K, const V3 (&sh)[NA], const V3 (&cand)[NB],
                                          unsigned alive) {
#pragma unroll 1
    for (int j = 0; j < K; j++) {
        const double *s = slot[j];
        const double a0 = s[0], a1 = s[1], a2 = s[2], b0 = s[3], b1 = s[4], b2 = s[5], g = s[6];
        unsigned hits = 0;
#pragma unroll
        for (int r = 0; r < NROUND; r++) {
            bool h;
            if (!ORCA) {
                const double vx = sh[r].x - a0, vy = sh[r].y - a1, vz = sh[r].z - a2;      // v_dif = (cand + pA) - apex
                const double dt = fma(b2, vz, fma(b1, vy, b0 * vx));
                const double n2 = fma(vz, vz, fma(vy, vy, vx * vx));
                h = (dt * fabs(dt) > g * n2) | (n2 == 0.0);
            } else {
                const double rx = cand[r].x - a0, ry = cand[r].y - a1, rz = cand[r].z - a2; // is_inORCA: (v - point) . n >= 0
                h = !(fma(rz, b2, fma(ry, b1, rx * b0)) >= 0.0);
            }
            hits |= (h ? 1u : 0u) << r;
        }
        alive &= ~hits;
        if (EARLY) { if (__ballot(alive != 0) == 0) break; }
    }
    return alive;
}

// The same pair test for a short tail of candidates (R <= 32): the wavefront is split into G = 64 / R' neighbour
// groups (R' = R rounded up to a power of two), lane = g * R' + t tests candidate t against neighbours g, g + G, ...;
// the caller ORs the groups' verdicts through one ballot.  K / G iterations instead of K.
template <bool ORCA, int SL>
__device__ __forceinline__ bool sweep_split(const double (*slot)[SL], int K, int G, int g, V3 sh, V3 cand) {
    bool hit = false;
    for (int j = g; j < K; j += G) {
        const double *s = slot[j];
        const double a0 = s[0], a1 = s[1], a2 = s[2], b0 = s[3], b1 = s[4], b2 = s[5], gg = s[6];
        bool h;
        if (!ORCA) {
            const double vx = sh.x - a0, vy = sh.y - a1, vz = sh.z - a2;
            const double dt = fma(b2, vz, fma(b1, vy, b0 * vx));
            const double n2 = fma(vz, vz, fma(vy, vy, vx * vx));
            h = (dt * fabs(dt) > gg * n2) | (n2 == 0.0);
        } else {
            const double rx = cand.x - a0, ry = cand.y - a1, rz = cand.z - a2;
            h = !(fma(rz, b2, fma(ry, b1, rx * b0)) >= 0.0);
        }
        hit |= h;
    }
    return hit;
}

__device__ __forceinline__ void solve_one(const DeviceView &d, const Params &Pctx, SolveLds &S, int agent, int lane, int wid) {
    const Params P = agent_params(d, Pctx, agent);        // (agent is wavefront-uniform: scalar loads)
    const PubRec me = d.rec[agent];
    int32_t *diag = d.diag + (size_t)agent * 8;
    if (me.flags & (FLAG_AT_GOAL | FLAG_COLLISION | FLAG_TIMEOUT)) {                // mampenv.py:35: row stays zero
        if (lane < 8) diag[lane] = -1;
        if (lane < 3) d.vpref_used[agent * 3 + lane] = __builtin_nan("");
        return;
    }
    const int pol = d.policy[agent];
    const bool orca = (pol == POL_ORCA || pol == POL_ORCA_LP);
    const V3 pA = v3(me.px, me.py, me.pz);
    F3 vA; vA.x = me.vx; vA.y = me.vy; vA.z = me.vz;
    const V3 vA64 = to_v3(vA);
    const double rA = me.radius;
    const double ps = d.pref_speed[agent];
    int st = 0;
    V3 vpref;
    if (d.vpref_mode[agent]) vpref = v3(d.vpref_ext[agent * 3], d.vpref_ext[agent * 3 + 1], d.vpref_ext[agent * 3 + 2]);
    else vpref = straight_v_pref(v3(d.goal[agent * 3], d.goal[agent * 3 + 1], d.goal[agent * 3 + 2]), pA, ps, orca);
    const bool first_step = l3norm_f32zero(vA, orca) <= 1e-5;                        // scaPolicy.py:34 / orca3dPolicy.py:53
    int dg_nsuit = -1, dg_fallback = -1, dg_chosen = -1, dg_pfail = -1, dg_lp4 = -1;
    V3 vpost;
    const int K = d.nbr_valid[agent] ? d.nbr_n[agent] : 0;
    if (first_step) {
        vpost = v3(0.3 * vpref.x, 0.3 * vpref.y, 0.3 * vpref.z);                     // scaPolicy.py:38
    } else {
        // ---- per-neighbour constants: lane j builds neighbour j, stages it in LDS -------------------
        double (*slot)[SLOT] = S.slot[wid];
        if (lane < K) {
            const int nid = d.nbr_id[agent * K_MAX + lane];
            V3 pB; F3 vB; double rB; bool stat; bool isob = (nid & NBR_OBSTACLE_BIT) != 0;
            if (isob) {
                const ObsRec o = d.obs[nid & ~NBR_OBSTACLE_BIT];
                pB = v3(o.px, o.py, o.pz); vB.x = vB.y = vB.z = 0.0f; rB = o.radius; stat = true;   // obstacle.py:22
            } else {
                const PubRec o = d.rec[nid];
                pB = v3(o.px, o.py, o.pz); vB.x = o.vx; vB.y = o.vy; vB.z = o.vz; rB = o.radius;
                stat = (o.flags & FLAG_AT_GOAL) != 0;
            }
            double *s = slot[lane];
            if (!orca) {
                const Cone c = make_cone(pA, vA, rA, pB, vB, rB, stat);
                s[0] = c.apex.x; s[1] = c.apex.y; s[2] = c.apex.z; s[3] = c.pAB.x; s[4] = c.pAB.y; s[5] = c.pAB.z;
                s[6] = c.g; s[7] = c.R; s[8] = c.absSq;
            } else {
                const OrcaOb o = make_orca(P, pA, vA, rA, pB, vB, rB, isob);
                s[0] = o.pl.p.x; s[1] = o.pl.p.y; s[2] = o.pl.p.z; s[3] = o.pl.n.x; s[4] = o.pl.n.y; s[5] = o.pl.n.z;
                s[6] = o.g; s[7] = o.R; s[8] = o.absSq; s[9] = o.relPos.x; s[10] = o.relPos.y; s[11] = o.relPos.z;
                const bool moving = o.vB_f32 ? (normf(o.vB) > (float)1e-5) : false;
                F3 h; h.x = 0.5f * (vA.x + vB.x); h.y = 0.5f * (vA.y + vB.y); h.z = 0.5f * (vA.z + vB.z);
                s[12] = moving ? (double)h.x : 0.0; s[13] = moving ? (double)h.y : 0.0; s[14] = moving ? (double)h.z : 0.0;
                Plane pl; pl.p = o.pl.p; pl.n = o.pl.n;
                S.planes[wid][lane] = pl;
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): LDS writes of this wave are visible to it
        if (pol == POL_ORCA_LP) {
            // ---- K3: LP3 (+LP4) -- orca3dPolicyOfficial.py:108-113.  Scalar chain, lane 0 drives.
            V3 nv = v3(0, 0, 0);
            int pf = 0, l4 = 0;
            if (lane == 0) {
                pf = lp3(S.planes[wid], K, P.max_speed, vpref, false, nv);
                if (pf < K) { lp4(S.planes[wid], K, pf, P.max_speed, nv, S.proj[wid]); l4 = 1; }
            }
            vpost = v3(__shfl(nv.x, 0), __shfl(nv.y, 0), __shfl(nv.z, 0));
            dg_pfail = __shfl(pf, 0);
            dg_lp4 = __shfl(l4, 0);
        } else {
            // ---- K2: candidate sweep ---------------------------------------------------------------
            CandTab T;
            T.num_N = (pol == POL_SCA && d.zaxis[agent]) ? 128 : 256;                 // scaPolicy.py:188-190
            T.unit = (T.num_N == 256) ? d.unit256 : d.unit128;
            T.phi = (T.num_N == 256) ? d.phi256 : d.phi128;
            T.vp_idx = 2 * T.num_N;
            if (!candidate_speeds(ps, T.rad1)) { st |= ST_BAD_PREF_SPEED; T.rad1 = ps; }
            const int nround = T.vp_idx >> 6;                                          // 8 or 4
            const double nvA = (double)normf(vA);
            V3 cand[NR + 1];
            V3 sh[NR];                                  // cand + pA, hoisted (same value as in the reference expression)
            int idx[NR + 1];
            unsigned okp = 0;                           // posture bits (util.py:6-20)
#pragma unroll
            for (int r = 0; r < NR; r++) {
                idx[r] = r * 64 + lane;
                if (r < nround) {
                    cand[r] = cand_from_idx(T, idx[r], vpref);
                    sh[r] = cand[r] + pA;
                    if (posture_ok(P, vA, nvA, pA.z, cand[r])) okp |= 1u << r;
                } else { cand[r] = v3(0, 0, 0); sh[r] = v3(0, 0, 0); }
            }
            cand[NR] = vpref;
            idx[NR] = T.vp_idx;
            const bool vp_post = posture_ok(P, vA, nvA, pA.z, vpref);
            // table candidates: neighbours outer (constants broadcast from LDS), candidates in registers
            unsigned alive;
            if (nround == NR) alive = orca ? sweep<NR, true, NR, NR + 1, SLOT>(slot, K, sh, cand, okp) : sweep<NR, false, NR, NR + 1, SLOT>(slot, K, sh, cand, okp);
            else alive = orca ? sweep<NR / 2, true, NR, NR + 1, SLOT>(slot, K, sh, cand, okp) : sweep<NR / 2, false, NR, NR + 1, SLOT>(slot, K, sh, cand, okp);
            // v_pref candidate: lane j tests neighbour j
            bool vp_hit = false;
            if (lane < K) {
                const double *s = slot[lane];
                if (!orca) {
                    Cone c; c.apex = v3(s[0], s[1], s[2]); c.pAB = v3(s[3], s[4], s[5]); c.g = s[6];
                    vp_hit = cone_hit(c, vpref + pA);
                } else {
                    Plane pl; pl.p = v3(s[0], s[1], s[2]); pl.n = v3(s[3], s[4], s[5]);
                    vp_hit = !in_orca(pl, vpref);
                }
            }
            const bool vp_ok = vp_post && (__ballot(vp_hit) == 0);
            int n_suit = vp_ok ? 1 : 0;
#pragma unroll
            for (int r = 0; r < NR; r++) n_suit += __popcll(__ballot((alive >> r) & 1u));
            dg_nsuit = n_suit;
            const bool shunted = (pol == POL_SCA || pol == POL_SRVO);
            bool inl[NR + 1];
            double key[NR + 1];
            int chosen;
            if (n_suit > 0) {
                dg_fallback = 0;
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    inl[r] = (alive >> r) & 1u;
                    key[r] = inl[r] ? l3norm(cand[r], vpref) : 0.0;                    // scaPolicy.py:219
                }
                inl[NR] = vp_ok && lane == 0;
                key[NR] = l3norm(vpref, vpref);
                chosen = select_from_list(shunted, pol == POL_SCA ? 3e-2 : 1e-1, n_suit, inl, key, idx, cand, T, vpref, vA64);
            } else {
                // ---- no suitable candidate: compute_without_suitV (scaPolicy.py:148-165,224-238) -----------
                dg_fallback = 1;
                double tcm[NR];
                bool have[NR];
#pragma unroll
                for (int r = 0; r < NR; r++) { tcm[r] = 0.0; have[r] = false; }
                for (int j = 0; j < K; j++) {
                    const double *s = slot[j];
#pragma unroll
                    for (int r = 0; r < NR; r++) {
                        if (r < nround && ((okp >> r) & 1u)) {
                            V3 pAB, vd; double g, R, absSq;
                            if (!orca) {
                                pAB = v3(s[3], s[4], s[5]); g = s[6]; R = s[7]; absSq = s[8];
                                vd = sh[r] - v3(s[0], s[1], s[2]);
                            } else {
                                pAB = v3(s[9], s[10], s[11]); g = s[6]; R = s[7]; absSq = s[8];
                                vd = cand[r] - v3(s[12], s[13], s[14]);                 // h == 0 when vB is static: v - 0 == v
                            }
                            if (cone_hit_vdif(pAB, g, vd)) {
                                const double tc = cone_tc(pAB, absSq, R, vd, &st);
                                if (!have[r] || tc < tcm[r]) { tcm[r] = tc; have[r] = true; }
                            }
                        }
                    }
                }
                // v_pref candidate, lanes over neighbours
                double tcv = INFINITY;
                if (lane < K && vp_post) {
                    const double *s = slot[lane];
                    V3 pAB, vd; double g, R, absSq;
                    if (!orca) { pAB = v3(s[3], s[4], s[5]); g = s[6]; R = s[7]; absSq = s[8]; vd = (vpref + pA) - v3(s[0], s[1], s[2]); }
                    else { pAB = v3(s[9], s[10], s[11]); g = s[6]; R = s[7]; absSq = s[8]; vd = vpref - v3(s[12], s[13], s[14]); }
                    if (cone_hit_vdif(pAB, g, vd)) tcv = cone_tc(pAB, absSq, R, vd, &st);
                }
                tcv = wave_min(tcv);
                if (tcv == INFINITY) tcv = 0.0;
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    inl[r] = r < nround;
                    key[r] = inl[r] ? (0.2 / (tcm[r] + 1e-5)) + l3norm(cand[r], vpref) : 0.0;   // scaPolicy.py:231-234
                }
                inl[NR] = lane == 0;
                key[NR] = (0.2 / (tcv + 1e-5)) + l3norm(vpref, vpref);
                chosen = select_from_list(shunted, pol == POL_SCA ? 5e-2 : 1e-1, T.vp_idx + 1, inl, key, idx, cand, T, vpref, vA64);
            }
            dg_chosen = chosen;
            vpost = trunc5(cand_from_idx(T, chosen, vpref));                          // scaPolicy.py:239
        }
    }
    // the per-agent scalar epilogue (cartesian2spherical, update_velocitie) runs one LANE per agent in k_action:
    // inside this one-wave-per-agent kernel it would cost a full wave's issue slots (~1500 instructions per agent)
    if (lane == 0) {
        d.vpost[agent * 3 + 0] = vpost.x; d.vpost[agent * 3 + 1] = vpost.y; d.vpost[agent * 3 + 2] = vpost.z;
        diag[0] = dg_nsuit; diag[1] = dg_fallback; diag[2] = dg_chosen; diag[3] = dg_pfail; diag[4] = dg_lp4;
        d.vpref_used[agent * 3 + 0] = vpref.x; d.vpref_used[agent * 3 + 1] = vpref.y; d.vpref_used[agent * 3 + 2] = vpref.z;
        const int stw = __builtin_amdgcn_readfirstlane(st);
        if (stw) atomicOr(&d.status[agent], stw);
    }
    // status bits raised by other lanes (fallback sqrt domain)
    if (lane != 0 && st) atomicOr(&d.status[agent], st);
}

// ------------------------------------------------------------------------------------------------
// K2 fast path.  The posture constraint (util.py:6-20: within max_heading_change of the current velocity) removes
// ~85 % of the 512 directions before any cone is looked at, so the candidates that pass it are COMPACTED across the
// wavefront (ballot + prefix into an LDS list) and only those are swept, two per lane at a time.  The survivors go to
// a second LDS list with their sort key; the selection passes run over that list.
// Integer numerator of l3norm(a, b) = round(sqrt(|a - b|^2), 5) (util.py:104) without the correctly rounded square root:
// rsq + one residual step gives sqrt to ~2^-46, i.e. the product with 1e5 to ~1e-9; unless that lands within 1e-6 of a
// rounding tie (checked for the whole wavefront) its nearest integer IS the reference's numerator.  Otherwise (about one
// wavefront in 4000) everybody takes the exact path.
__device__ __forceinline__ double l3norm_num(V3 a, V3 b) {
    const double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    double s = dx * dx + dy * dy;
    s = s + dz * dz;
    const double h = __builtin_amdgcn_rsq(s);
    const double q0 = s * h;
    const double q1 = fma(0.5 * fma(-q0, q0, s), h, q0);
    const double y = q1 * EPS5;
    double r = rint(y);
    const bool unsure = !(fabs(fabs(y - r) - 0.5) > 1e-6) || !(s > 1e-200);        // near a tie, zero, or not finite
    if (__any(unsure)) {
        double k;
        round5_py(sqrt(s), &k);
        r = k;
    }
    return r;
}

// self-test of l3norm_num against the exact path (tests/test_gpu_parity.py feeds random and near-tie inputs)
__global__ __launch_bounds__(64) void k_selftest_l3norm(const double *a, const double *b, int n, double *fast, double *exact) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = i < n ? i : n - 1;
    const V3 va = v3(a[3 * j], a[3 * j + 1], a[3 * j + 2]), vb = v3(b[3 * j], b[3 * j + 1], b[3 * j + 2]);
    const double f = l3norm_num(va, vb);
    double k;
    l3norm(va, vb, &k);
    if (i < n) { fast[i] = f; exact[i] = k; }
}

struct FastLds {
    double slot[SOLVE_WAVES][K_MAX][SLOTF];
    union PerWave {                                // the LP policy never builds candidate lists
        struct { Plane planes[K_MAX]; Plane proj[K_MAX]; } lp;
        struct { unsigned int pkS[520];            // survivors: (round5 numerator of |v - v_pref|) << 10 | generation index
                 unsigned short listA[512]; } cl;
    } u[SOLVE_WAVES];
};

// PHASE 0: the whole of it in one launch (k_solve).  Nothing up to the list of table candidates that survive the cones depends
// on v_pref -- the cones, the posture filter and the sweep read positions, velocities and the static tables only -- so a pass
// whose v_pref comes from the device tracker runs that part (PHASE 1, k_solve_sweep) beside the re-plans, on the stream of
// the neighbour query, and only the rest behind them (k_solve_pick4 / solve_pick4 below: distances to v_pref, the v_pref
// candidate itself, the selection).  Phase 1 leaves the cones and the survivors' generation indices in global memory; the
// same expressions on the same values either way, so the split changes no bit.
// LPMODE 1: compiled without the LP chain (the scalar LP1-4 on lane 0 cost every other policy 22 registers and the spills of
// a 7-waves-per-SIMD build): an ORCA3D-Official agent past its bootstrap step is left to k_solve_lpw / k_lp, everything else
// about it (done flags, bootstrap velocity) is still handled here.  LPMODE 2 (k_solve_lpw): those agents only, after k_solve.
template <int PHASE, int LPMODE>
__device__ __forceinline__ void solve_fast(const DeviceView &d, const Params &Pctx, FastLds &S, int agent, int lane, int wid) {
    const Params P = agent_params(d, Pctx, agent);        // (agent is wavefront-uniform: scalar loads)
    const PubRec me = d.rec[agent];
    int32_t *diag = d.diag + (size_t)agent * 8;
    if (PHASE != 1 && LPMODE != 2 && lane == 0) d.is_fb[agent] = 0;
    if (me.flags & (FLAG_AT_GOAL | FLAG_COLLISION | FLAG_TIMEOUT)) {                // mampenv.py:35
        if (PHASE == 1 || LPMODE == 2) return;
        if (lane < 8) diag[lane] = -1;
        if (lane < 3) d.vpref_used[agent * 3 + lane] = __builtin_nan("");
        return;
    }
    const int pol = d.policy[agent];
    const bool orca = (pol == POL_ORCA || pol == POL_ORCA_LP);
    const V3 pA = v3(me.px, me.py, me.pz);
    F3 vA; vA.x = me.vx; vA.y = me.vy; vA.z = me.vz;
    const V3 vA64 = to_v3(vA);
    const double rA = me.radius;
    Prep pr;                                              // per-agent scalar prologue (prep_agent)
    if (PHASE == 1) {
        // the prologue of this pass is written behind the re-plans (it holds v_pref); what the sweep reads of it, recomputed
        // by the same expressions (prep_agent): bootstrap test, |vA|, the second candidate speed
        pr.vpref[0] = pr.vpref[1] = pr.vpref[2] = 0.0;
        pr.nvA = (double)normf(vA);
        pr.bits = (l3norm_f32zero(vA, orca) <= 1e-5) ? 1u : 0u;
        double rad1;
        if (!candidate_speeds(d.pref_speed[agent], rad1)) rad1 = d.pref_speed[agent];
        pr.rad1 = rad1;
        pr.vp_key = 0;
        if (pol == POL_ORCA_LP || (pr.bits & 1u)) return;                            // no candidate sweep for these
    } else pr = ((const Prep *)d.prep)[agent];
    int st = ((pr.bits & 2u) ? ST_BAD_PREF_SPEED : 0);
    const V3 vpref = v3(pr.vpref[0], pr.vpref[1], pr.vpref[2]);
    const bool first_step = (pr.bits & 1u) != 0;
    int dg_nsuit = -1, dg_fallback = -1, dg_chosen = -1, dg_pfail = -1, dg_lp4 = -1;
    V3 vpost = v3(0, 0, 0);
    bool defer = false;
    const int K = d.nbr_valid[agent] ? d.nbr_n[agent] : 0;
    if (pol == POL_ORCA_LP && !first_step && (LPMODE == 1 || d.lp_kernel)) return;   // K3: k_lp (one lane per agent) or k_solve_lpw
    if (LPMODE == 2 && first_step) return;                                           // k_solve has done the bootstrap step
    if (first_step) {
        vpost = v3(0.3 * vpref.x, 0.3 * vpref.y, 0.3 * vpref.z);                     // scaPolicy.py:38
    } else {
        double (*slot)[SLOTF] = S.slot[wid];
        if (lane < K) {                                   // lane j builds neighbour j (scaPolicy.py:47-60 / orca :57-107)
            const int nid = d.nbr_id[agent * K_MAX + lane];
            V3 pB; F3 vB; double rB; bool stat; const bool isob = (nid & NBR_OBSTACLE_BIT) != 0;
            if (isob) {
                const ObsRec o = d.obs[nid & ~NBR_OBSTACLE_BIT];
                pB = v3(o.px, o.py, o.pz); vB.x = vB.y = vB.z = 0.0f; rB = o.radius; stat = true;   // obstacle.py:22
            } else {
                const PubRec o = d.rec[nid];
                pB = v3(o.px, o.py, o.pz); vB.x = o.vx; vB.y = o.vy; vB.z = o.vz; rB = o.radius;
                stat = (o.flags & FLAG_AT_GOAL) != 0;
            }
            double *sl = slot[lane];
            if (!orca) {
                const Cone c = make_cone(pA, vA, rA, pB, vB, rB, stat);
                sl[0] = c.apex.x; sl[1] = c.apex.y; sl[2] = c.apex.z; sl[3] = c.pAB.x; sl[4] = c.pAB.y; sl[5] = c.pAB.z; sl[6] = c.g;
            } else {
                const OrcaOb o = make_orca(P, pA, vA, rA, pB, vB, rB, isob);
                sl[0] = o.pl.p.x; sl[1] = o.pl.p.y; sl[2] = o.pl.p.z; sl[3] = o.pl.n.x; sl[4] = o.pl.n.y; sl[5] = o.pl.n.z; sl[6] = 0.0;
                Plane pl; pl.p = o.pl.p; pl.n = o.pl.n;
                if (pol == POL_ORCA_LP) S.u[wid].lp.planes[lane] = pl;
            }
            if (PHASE == 1) {
                double *g = d.sw_slot + (size_t)agent * K_MAX * SLOTF + lane;
#pragma unroll
                for (int q = 0; q < 7; q++) g[q * K_MAX] = sl[q];
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (LPMODE != 1 && pol == POL_ORCA_LP) {
            // K3, wave-per-agent form (shards with few LP agents: launch_policy picks): LP3 (+LP4), orca3dPolicyOfficial.py:108-113.
            // Scalar chain, lane 0 drives: measured faster than the lanes-over-planes
            // form (ballots for the next violated plane, wave max / min for LP1) at every BASELINE size but N = 1024, DESIGN.md section 3.
            V3 nv = v3(0, 0, 0);
            int pf = 0, l4 = 0;
            if (lane == 0) {
                pf = lp3(S.u[wid].lp.planes, K, P.max_speed, vpref, false, nv);
                if (pf < K) { lp4(S.u[wid].lp.planes, K, pf, P.max_speed, nv, S.u[wid].lp.proj); l4 = 1; }
            }
            vpost = v3(__shfl(nv.x, 0), __shfl(nv.y, 0), __shfl(nv.z, 0));
            dg_pfail = __shfl(pf, 0);
            dg_lp4 = __shfl(l4, 0);
        } else if (LPMODE != 2) {
            CandTab T;
            T.num_N = (pol == POL_SCA && d.zaxis[agent]) ? 128 : 256;                 // scaPolicy.py:188-190
            T.unit = (T.num_N == 256) ? d.unit256 : d.unit128;
            T.phi = (T.num_N == 256) ? d.phi256 : d.phi128;
            T.vp_idx = 2 * T.num_N;
            T.rad1 = pr.rad1;
            const double nvA = pr.nvA;
            unsigned short *listA = S.u[wid].cl.listA;
            unsigned int *pkS = S.u[wid].cl.pkS;
            // ---- posture filter (util.py:6-20) + compaction.  c >= thr is decided without sqrt / division whenever
            //      dot^2 and (thr*|vA|)^2 |v|^2 are more than 1e-13 apart (relative); otherwise the exact expression runs.
            const double thr = P.cos_heading_thr;
            const double tn = thr * nvA;
            const double T2 = tn * tn;
            // Both speeds of a direction share the verdict whenever it is "sure": c = dot / (|vA||v|) does not depend on the
            // candidate's length beyond rounding (<= 1e-15 relative here, the margin is 1e-13); the z floor is per speed.
            int nA = 0, nS = 0;
            const bool filter_ok = thr > 1e-6;
            const int ndir = T.num_N >> 6;
            // util.py:16 `pos.z + dt * v.z >= 0` holds for every candidate once the agent is higher than the fastest one can sink
            const bool z_safe = pA.z > P.time_step * (T.rad1 > 0.5 ? T.rad1 : 0.5) * 1.000001;
            for (int r = 0; r < ndir; r++) {
                const int n0 = r * 64 + lane;
                const double ux = T.unit[n0], uy = T.unit[T.num_N + n0], uz = T.unit[2 * T.num_N + n0];
                const V3 u = v3(ux, uy, uz);
                const double dt = dot(vA64, u);
                const double y = T2;                       // T2 * |u|^2 with |u|^2 = 1 +- 4e-16 (table of unit vectors): inside the margin
                const double lhs = dt * dt;
                const bool sure_pass = filter_ok & (dt > 0.0) & (lhs > y * (1.0 + 1e-13));
                const bool sure_fail = filter_ok & ((dt <= 0.0) | (lhs < y * (1.0 - 1e-13)));
                const bool unsure = !(sure_pass | sure_fail);
                const bool any_unsure = __ballot(unsure) != 0;
#pragma unroll
                for (int sp = 0; sp < 2; sp++) {
                    const double rad = sp ? T.rad1 : 0.5;
                    const V3 c = v3(rad * ux, rad * uy, rad * uz);
                    bool ok = sure_pass;
                    if (any_unsure) { if (unsure) ok = posture_cos(vA, nvA, c) >= thr; }
                    if (!z_safe) ok = ok & ((pA.z + P.time_step * c.z) >= 0.0);
                    const unsigned long long m = __ballot(ok);
                    if (ok) listA[nA + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(sp * T.num_N + n0);
                    nA += __popcll(m);
                }
            }
            __builtin_amdgcn_wave_barrier();
            // ---- sweep of the compacted candidates; survivors -> packed keys.  Full groups of 128: two candidates per lane;
            //      then one group of 64; a tail of <= 32 candidates splits the neighbours across lane groups instead.
            uint16_t *surv = PHASE == 1 ? d.sw_surv + (size_t)agent * 512 : nullptr;
            auto emit = [&](bool a, V3 cdq, int ixq) {
                const unsigned long long m = __ballot(a);
                if (a) {
                    const int at = nS + __popcll(m & ((1ull << lane) - 1ull));
                    if (PHASE == 1) surv[at] = (uint16_t)ixq;
                    else pkS[at] = pack_key(l3norm_num(cdq, vpref), ixq);                                       // scaPolicy.py:219
                }
                nS += __popcll(m);
            };
            int c0 = 0;
            for (; nA - c0 >= 128; c0 += 128) {
                V3 cd[2], sh[2];
                int ix[2];
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    ix[q] = (int)listA[c0 + q * 64 + lane];
                    cd[q] = cand_from_idx(T, ix[q], vpref);
                    sh[q] = cd[q] + pA;
                }
                const unsigned alive = orca ? sweep<2, true, 2, 2, SLOTF, false>(slot, K, sh, cd, 3u) : sweep<2, false, 2, 2, SLOTF, false>(slot, K, sh, cd, 3u);
#pragma unroll
                for (int q = 0; q < 2; q++) emit((alive >> q) & 1u, cd[q], ix[q]);
            }
            while (c0 < nA) {
                const int rem = nA - c0;
                if (rem > 32) {
                    const int cnt = rem < 64 ? rem : 64;
                    V3 cd[1], sh[1];
                    const int ix = lane < cnt ? (int)listA[c0 + lane] : 0;
                    cd[0] = cand_from_idx(T, ix, vpref);
                    sh[0] = cd[0] + pA;
                    const unsigned valid = lane < cnt ? 1u : 0u;
                    const unsigned alive = orca ? sweep<1, true, 1, 1, SLOTF, false>(slot, K, sh, cd, valid) : sweep<1, false, 1, 1, SLOTF, false>(slot, K, sh, cd, valid);
                    emit(alive & 1u, cd[0], ix);
                    c0 += cnt;
                } else {
                    int Rp = 1;
                    while (Rp < rem) Rp <<= 1;                                           // 1 .. 32
                    const int G = 64 / Rp;
                    const int t = lane & (Rp - 1), g = lane / Rp;
                    const int ix = t < rem ? (int)listA[c0 + t] : 0;
                    const V3 cd = cand_from_idx(T, ix, vpref);
                    const bool hit = orca ? sweep_split<true, SLOTF>(slot, K, G, g, cd + pA, cd) : sweep_split<false, SLOTF>(slot, K, G, g, cd + pA, cd);
                    unsigned long long m = __ballot(hit);
                    for (int w = 32; w >= Rp; w >>= 1) m |= m >> w;                       // OR over the neighbour groups
                    emit((lane < rem) & !((m >> t) & 1ull), cd, ix);
                    c0 += rem;
                }
            }
            if (PHASE == 1) { if (lane == 0) d.sw_n[agent] = nS; return; }
            // ---- the v_pref candidate (scaPolicy.py:206-211): lane j tests neighbour j
            bool vp_hit = false;
            if (lane < K) {
                const double *sl = slot[lane];
                if (!orca) {
                    Cone c; c.apex = v3(sl[0], sl[1], sl[2]); c.pAB = v3(sl[3], sl[4], sl[5]); c.g = sl[6];
                    vp_hit = cone_hit(c, vpref + pA);
                } else {
                    Plane pl; pl.p = v3(sl[0], sl[1], sl[2]); pl.n = v3(sl[3], sl[4], sl[5]);
                    vp_hit = !in_orca(pl, vpref);
                }
            }
            const bool vp_ok = (pr.bits & 4u) && (__ballot(vp_hit) == 0);
            if (vp_ok) {
                if (lane == 0) pkS[nS] = pr.vp_key | (unsigned)T.vp_idx;
                nS++;
            }
            __builtin_amdgcn_wave_barrier();
            dg_nsuit = nS;
            if (nS == 0) {
                // no suitable candidate: compute_without_suitV needs all 513 candidates -> k_solve_full finishes this agent
                if (lane == 0) { const int at = atomicAdd(d.fb_count, 1); d.fb_list[at] = agent; d.is_fb[agent] = 1; }
                return;
            }
            dg_fallback = 0;
            // ---- selection over the survivor list.  The reference's stable sort by round5(|v - v_pref|) (scaPolicy.py:219)
            //      is the order of the packed integer (numerator << 10 | generation index): one 32-bit wave reduction each.
            const bool shunted = (pol == POL_SCA || pol == POL_SRVO);
            unsigned best = 0xffffffffu;
            for (int e = lane; e < nS; e += 64) { const unsigned k = pkS[e]; best = k < best ? k : best; }
            best = wave_min_u32(best);
            int chosen = (int)(best & 1023u);
            if (shunted && nS > 1) {                                                     // scaPolicy.py:119-145
                // |l3norm(v0, vA) - l3norm(vi, vA)| < thr on the rounded values: both are k / 1e5 with integer k, so the
                // verdict is the integer comparison |k0 - ki| vs thr * 1e5 except when they are equal (then the doubles decide)
                const double sthr = pol == POL_SCA ? 3e-2 : 1e-1;
                const double kthr = pol == POL_SCA ? 3000.0 : 10000.0;
                // numerators of l3norm(v_i, vA) for the list (entries lane and lane + 64 in registers; longer lists loop);
                // the best entry's numerator k0 is picked up from the lane that holds it instead of being computed again
                const unsigned kA = lane < nS ? pkS[lane] : 0xffffffffu;
                const unsigned kB = lane + 64 < nS ? pkS[lane + 64] : 0xffffffffu;
                double nA_ = 0.0, nB_ = 0.0;
                if (lane < nS) nA_ = l3norm_num(cand_from_idx(T, (int)(kA & 1023u), vpref), vA64);
                if (nS > 64) { if (lane + 64 < nS) nB_ = l3norm_num(cand_from_idx(T, (int)(kB & 1023u), vpref), vA64); }
                double k0;
                {
                    const unsigned long long mA = __ballot(kA == best), mB = __ballot(kB == best);
                    if (mA != 0) k0 = readlane_f64(nA_, __ffsll((long long)mA) - 1);
                    else if (mB != 0) k0 = readlane_f64(nB_, __ffsll((long long)mB) - 1);
                    else k0 = l3norm_num(cand_from_idx(T, chosen, vpref), vA64);         // best sits beyond entry 127
                }
                auto passes = [&](double kv) {
                    const double dk = fabs(k0 - kv);
                    bool pass = dk < kthr;
                    if (dk == kthr) pass = fabs(k0 / EPS5 - kv / EPS5) < sthr;          // round5_py returns k / 1e5
                    return pass;
                };
                unsigned fail = 0xffffffffu;                                             // first list element that breaks the prefix
                if (lane < nS && !passes(nA_)) fail = kA;
                if (lane + 64 < nS && !passes(nB_)) fail = kB < fail ? kB : fail;
                for (int e = lane + 128; e < nS; e += 64) {
                    const unsigned k = pkS[e];
                    if (!passes(l3norm_num(cand_from_idx(T, (int)(k & 1023u), vpref), vA64))) fail = k < fail ? k : fail;
                }
                fail = wave_min_u32(fail);
                // first minimal / first maximal get_phi inside the prefix, "first" in list order (= order of the packed keys):
                // per lane the best (phi, key) pair, then two 32-bit wave minima each (phi, then the key among the lanes that hold it)
                unsigned pmin = 0xffffffffu, kpmin = 0xffffffffu, pmax = 0xffffffffu, kpmax = 0xffffffffu;   // pmax holds 0xfffff - phi
                auto consider = [&](unsigned k) {
                    if (k < fail) {                                                      // 0xffffffff (no entry) never is
                        const int ci = (int)(k & 1023u);
                        const unsigned ph = ci >= T.vp_idx ? (pr.bits >> 8) : (unsigned)T.phi[ci >= T.num_N ? ci - T.num_N : ci];   // <= 628318
                        const unsigned ih = 0xfffffu - ph;
                        if (ph < pmin || (ph == pmin && k < kpmin)) { pmin = ph; kpmin = k; }
                        if (ih < pmax || (ih == pmax && k < kpmax)) { pmax = ih; kpmax = k; }
                    }
                };
                consider(kA);
                if (nS > 64) consider(kB);
                for (int e = lane + 128; e < nS; e += 64) consider(pkS[e]);
                const unsigned wpmin = wave_min_u32(pmin), wpmax = wave_min_u32(pmax);
                const unsigned kmin = wave_min_u32(pmin == wpmin ? kpmin : 0xffffffffu);
                const unsigned kmax = wave_min_u32(pmax == wpmax ? kpmax : 0xffffffffu);
                const double phi_min = (double)wpmin / EPS5, phi_max = (double)(0xfffffu - wpmax) / EPS5;
                chosen = (fabs(phi_max - phi_min) <= PI) ? (int)(kmin & 1023u) : (int)(kmax & 1023u);
            }
            dg_chosen = chosen;
            defer = true;                         // vA_post = trunc5(candidate `chosen`) (scaPolicy.py:239): the epilogue derives it
        }
    }
    if (lane == 0) {
        if (defer) d.is_fb[agent] = 2;
        else { d.vpost[agent * 3 + 0] = vpost.x; d.vpost[agent * 3 + 1] = vpost.y; d.vpost[agent * 3 + 2] = vpost.z; }
        diag[0] = dg_nsuit; diag[1] = dg_fallback; diag[2] = dg_chosen; diag[3] = dg_pfail; diag[4] = dg_lp4;
        d.vpref_used[agent * 3 + 0] = vpref.x; d.vpref_used[agent * 3 + 1] = vpref.y; d.vpref_used[agent * 3 + 2] = vpref.z;
        if (st) atomicOr(&d.status[agent], st);
    }
}

__global__ __launch_bounds__(SOLVE_WAVES * 64, 8) void k_solve(DeviceView d, Params P) {
    SCA_TL(d, TL_SOLVE);
    __shared__ FastLds S;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // one agent per wavefront, no grid-stride loop (the loop form costs registers: 1 wave/SIMD instead of 2)
    const int idx = blockIdx.x * SOLVE_WAVES + wid;
    if (idx < shard_size(d)) solve_fast<0, 1>(d, P, S, shard_agent(d, idx), lane, wid);
}
// K3, wave-per-agent form: the ORCA3D-Official agents of the shard (positions [lo, hi) of the sorted list of their ids) when
// they are too few for k_lp to fill the chip
__global__ __launch_bounds__(SOLVE_WAVES * 64) void k_solve_lpw(DeviceView d, Params P, const int32_t *list, int lo, int hi) {
    SCA_TL(d, TL_LP);
    __shared__ FastLds S;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int idx = lo + blockIdx.x * SOLVE_WAVES + wid;
    // (partition mode hands over all owned agents: `hi` is then the host's bound and the exact count is on the device)
    if (idx < hi && !(d.count_dev && idx >= shard_size(d)) && d.policy[list[idx]] == POL_ORCA_LP) solve_fast<0, 2>(d, P, S, list[idx], lane, wid);
}
// the first half of k_solve for passes whose v_pref arrives late (solve_fast); the second is k_solve_pick4
__global__ __launch_bounds__(SOLVE_WAVES * 64) void k_solve_sweep(DeviceView d, Params P) {
    SCA_TL(d, TL_SOLVE_SWEEP);
#if defined(SCA_SWEEP_PRIO) && SCA_SWEEP_PRIO > 0
    __builtin_amdgcn_s_setprio(SCA_SWEEP_PRIO);
#endif
    __shared__ FastLds S;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int idx = blockIdx.x * SOLVE_WAVES + wid;
    if (idx < shard_size(d)) solve_fast<1, 1>(d, P, S, shard_agent(d, idx), lane, wid);
}


// ------------------------------------------------------------------------------------------------
// k_solve_pick4, FOUR AGENTS PER WAVEFRONT.  What is left of find_next_action behind the sweep is short per agent -- a few
// dozen survivors, <= 16 neighbours for the v_pref candidate -- and a wavefront per agent spends most of its instructions on
// 64-lane reductions and per-agent scalars.  Here an agent owns one 16-lane DPP row: lane j of the row tests neighbour j and
// holds survivors j, j + 16, ...; every reduction is four DPP steps inside the row; rows finish independently.  Same
// expressions as solve_fast<0> (the selection is order-free: the packed keys carry the generation index), so the result is
// the one-launch kernel's bit for bit (tests/test_gpu_solve_split.py).
constexpr int PICK_APW = 4;
constexpr int PICK_CAP = 144;             // packed keys per agent cached in LDS (9 per lane); longer lists recompute the rest
struct PickLds { unsigned int pk[SOLVE_WAVES][PICK_APW][PICK_CAP]; };

__device__ __forceinline__ unsigned row_min_u32(unsigned x) {             // every lane of the row gets the row's minimum
    int v = (int)x;
    v = (int)umin32((unsigned)v, (unsigned)dpp_mov<0xb1, 0xf>(v));        // quad_perm [1,0,3,2]
    v = (int)umin32((unsigned)v, (unsigned)dpp_mov<0x4e, 0xf>(v));        // quad_perm [2,3,0,1]
    v = (int)umin32((unsigned)v, (unsigned)dpp_mov<0x124, 0xf>(v));       // row_ror:4
    v = (int)umin32((unsigned)v, (unsigned)dpp_mov<0x128, 0xf>(v));       // row_ror:8
    return (unsigned)v;
}

__device__ __forceinline__ void solve_pick4(const DeviceView &d, const Params &P, unsigned int *pk, int agent, int j, int row) {
    // Everything the row reads that is addressed by the agent id alone is loaded up front, before the first branch: the kernel is
    // a chain of dependent memory round trips (record -> prologue -> cones / survivors), not arithmetic, and the loads below the
    // early exits would each wait for the previous one's data (45 -> 41 us at c4).  All addresses are valid for any agent.
    const PubRec me = d.rec[agent];
    const int pol_ = d.policy[agent];
    const Prep pr_ = ((const Prep *)d.prep)[agent];
    const int nbrv_ = d.nbr_valid[agent], nbrn_ = d.nbr_n[agent], zax_ = d.zaxis[agent], nT_ = d.sw_n[agent];
    double sl_[7];
    {
        const double *g = d.sw_slot + (size_t)agent * K_MAX * SLOTF + j;
#pragma unroll
        for (int q = 0; q < 7; q++) sl_[q] = g[q * K_MAX];
    }
    int32_t *diag = d.diag + (size_t)agent * 8;
    if (j == 0) d.is_fb[agent] = 0;
    if (me.flags & (FLAG_AT_GOAL | FLAG_COLLISION | FLAG_TIMEOUT)) {                // mampenv.py:35
        if (j < 8) diag[j] = -1;
        if (j < 3) d.vpref_used[agent * 3 + j] = __builtin_nan("");
        return;
    }
    const int pol = pol_;
    const bool orca = (pol == POL_ORCA || pol == POL_ORCA_LP);
    const V3 pA = v3(me.px, me.py, me.pz);
    F3 vA; vA.x = me.vx; vA.y = me.vy; vA.z = me.vz;
    const V3 vA64 = to_v3(vA);
    const Prep pr = pr_;
    const int st = ((pr.bits & 2u) ? ST_BAD_PREF_SPEED : 0);
    const V3 vpref = v3(pr.vpref[0], pr.vpref[1], pr.vpref[2]);
    const bool first_step = (pr.bits & 1u) != 0;
    int dg_nsuit = -1, dg_fallback = -1, dg_chosen = -1;
    V3 vpost = v3(0, 0, 0);
    bool defer = false;
    const int K = nbrv_ ? nbrn_ : 0;
    if (pol == POL_ORCA_LP && !first_step) return;                                   // K3: k_lp
    if (first_step) {
        vpost = v3(0.3 * vpref.x, 0.3 * vpref.y, 0.3 * vpref.z);                     // scaPolicy.py:38
    } else {
        CandTab T;
        T.num_N = (pol == POL_SCA && zax_) ? 128 : 256;                               // scaPolicy.py:188-190
        T.unit = (T.num_N == 256) ? d.unit256 : d.unit128;
        T.phi = (T.num_N == 256) ? d.phi256 : d.phi128;
        T.vp_idx = 2 * T.num_N;
        T.rad1 = pr.rad1;
        // ---- the v_pref candidate (scaPolicy.py:206-211): lane j tests neighbour j against the cone / plane of the sweep
        bool vp_hit = false;
        if (j < K) {
            if (!orca) {
                Cone c; c.apex = v3(sl_[0], sl_[1], sl_[2]); c.pAB = v3(sl_[3], sl_[4], sl_[5]); c.g = sl_[6];
                vp_hit = cone_hit(c, vpref + pA);
            } else {
                Plane pl; pl.p = v3(sl_[0], sl_[1], sl_[2]); pl.n = v3(sl_[3], sl_[4], sl_[5]);
                vp_hit = !in_orca(pl, vpref);
            }
        }
        const unsigned long long hits = __ballot(vp_hit);
        const bool vp_ok = (pr.bits & 4u) && (((hits >> (16 * row)) & 0xffffull) == 0);
        // ---- packed keys of the survivors: (round5 numerator of |v - v_pref|) << 10 | generation index (scaPolicy.py:219)
        const int nT = nT_;
        const uint16_t *surv = d.sw_surv + (size_t)agent * 512;
        auto key_of = [&](int e) {
            const int ix = (int)surv[e];
            return pack_key(l3norm_num(cand_from_idx(T, ix, vpref), vpref), ix);
        };
        unsigned best = 0xffffffffu;
        for (int e = j; e < nT; e += 16) {
            const unsigned k = key_of(e);
            if (e < PICK_CAP) pk[e] = k;
            best = k < best ? k : best;
        }
        auto entry = [&](int e) { return e < PICK_CAP ? pk[e] : key_of(e); };       // a lane re-reads only what it wrote itself
        const unsigned vpk = vp_ok ? (pr.vp_key | (unsigned)T.vp_idx) : 0xffffffffu;
        const int nS = nT + (vp_ok ? 1 : 0);
        dg_nsuit = nS;
        if (nS == 0) {
            // no suitable candidate: compute_without_suitV needs all 513 candidates -> the epilogue's second half finishes this agent
            if (j == 0) { const int at = atomicAdd(d.fb_count, 1); d.fb_list[at] = agent; d.is_fb[agent] = 1; }
            return;
        }
        dg_fallback = 0;
        best = vpk < best ? vpk : best;
        best = row_min_u32(best);
        int chosen = (int)(best & 1023u);
        const bool shunted = (pol == POL_SCA || pol == POL_SRVO);
        if (shunted && nS > 1) {                                                         // scaPolicy.py:119-145, as in solve_fast
            const double sthr = pol == POL_SCA ? 3e-2 : 1e-1;
            const double kthr = pol == POL_SCA ? 3000.0 : 10000.0;
            const double k0 = l3norm_num(cand_from_idx(T, chosen, vpref), vA64);
            auto passes = [&](double kv) {
                const double dk = fabs(k0 - kv);
                bool pass = dk < kthr;
                if (dk == kthr) pass = fabs(k0 / EPS5 - kv / EPS5) < sthr;
                return pass;
            };
            unsigned fail = 0xffffffffu;                                                 // first list element that breaks the prefix
            for (int e = j; e < nT; e += 16) {
                const unsigned k = entry(e);
                if (!passes(l3norm_num(cand_from_idx(T, (int)(k & 1023u), vpref), vA64))) fail = k < fail ? k : fail;
            }
            if (vp_ok) { if (!passes(l3norm_num(vpref, vA64))) fail = vpk < fail ? vpk : fail; }
            fail = row_min_u32(fail);
            unsigned pmin = 0xffffffffu, kpmin = 0xffffffffu, pmax = 0xffffffffu, kpmax = 0xffffffffu;   // pmax holds 0xfffff - phi
            auto consider = [&](unsigned k) {
                if (k < fail) {
                    const int ci = (int)(k & 1023u);
                    const unsigned ph = ci >= T.vp_idx ? (pr.bits >> 8) : (unsigned)T.phi[ci >= T.num_N ? ci - T.num_N : ci];
                    const unsigned ih = 0xfffffu - ph;
                    if (ph < pmin || (ph == pmin && k < kpmin)) { pmin = ph; kpmin = k; }
                    if (ih < pmax || (ih == pmax && k < kpmax)) { pmax = ih; kpmax = k; }
                }
            };
            for (int e = j; e < nT; e += 16) consider(entry(e));
            consider(vpk);
            const unsigned wpmin = row_min_u32(pmin), wpmax = row_min_u32(pmax);
            const unsigned kmin = row_min_u32(pmin == wpmin ? kpmin : 0xffffffffu);
            const unsigned kmax = row_min_u32(pmax == wpmax ? kpmax : 0xffffffffu);
            const double phi_min = (double)wpmin / EPS5, phi_max = (double)(0xfffffu - wpmax) / EPS5;
            chosen = (fabs(phi_max - phi_min) <= PI) ? (int)(kmin & 1023u) : (int)(kmax & 1023u);
        }
        dg_chosen = chosen;
        defer = true;
    }
    if (j == 0) {
        if (defer) d.is_fb[agent] = 2;
        else { d.vpost[agent * 3 + 0] = vpost.x; d.vpost[agent * 3 + 1] = vpost.y; d.vpost[agent * 3 + 2] = vpost.z; }
        diag[0] = dg_nsuit; diag[1] = dg_fallback; diag[2] = dg_chosen; diag[3] = -1; diag[4] = -1;
        d.vpref_used[agent * 3 + 0] = vpref.x; d.vpref_used[agent * 3 + 1] = vpref.y; d.vpref_used[agent * 3 + 2] = vpref.z;
        if (st) atomicOr(&d.status[agent], st);
    }
}

__global__ __launch_bounds__(SOLVE_WAVES * 64) void k_solve_pick4(DeviceView d, Params P) {
    SCA_TL(d, TL_SOLVE_PICK);
    __shared__ PickLds S;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = lane >> 4;
    const int idx = (blockIdx.x * SOLVE_WAVES + wid) * PICK_APW + row;
    if (idx < shard_size(d)) solve_pick4(d, P, S.pk[wid][row], shard_agent(d, idx), lane & 15, row);   // whole rows leave together
}

// ------------------------------------------------------------------------------------------------
// K3: the "Official" ORCA3D policy (orca3dPolicyOfficial.py:56-113), ONE LANE PER AGENT.  The LP is a sequential walk over
// <= 16 planes with data-dependent recursion (LP3 -> LP2 -> LP1): inside the one-wavefront-per-agent k_solve it ran on lane 0
// with 63 lanes idle; here 64 agents share a wavefront, each lane builds its agent's planes (make_orca per neighbour) into a
// lane-transposed LDS array -- plane j, component c of lane l at [j][c][l]: every access is 64 consecutive doubles -- and
// walks them with the scalar LP of sca_core.h through an accessor.  Same statements, same order: bit for bit the planes and
// velocities of the wave-per-agent form.  Agents whose LP3 fails (planeFail < K) need LP4's projected planes: they go to
// the fallback list and are finished, one wavefront each, by k_fallback (solve_one), like the agents
// without a suitable candidate.
struct LpPlanes { double v[K_MAX][6][64]; };                       // 48 KB per wavefront
struct LpAccess {
    const LpPlanes *S; int lane;
    __device__ __forceinline__ Plane operator[](int i) const {
        Plane q;
        q.p = v3(S->v[i][0][lane], S->v[i][1][lane], S->v[i][2][lane]);
        q.n = v3(S->v[i][3][lane], S->v[i][4][lane], S->v[i][5][lane]);
        return q;
    }
};
__global__ __launch_bounds__(64) void k_lp(DeviceView d, Params Pctx, const int32_t *list, int lo, int hi) {
    SCA_TL(d, TL_LP);
    __shared__ LpPlanes S;
    const int lane = threadIdx.x;
    const int at = lo + blockIdx.x * 64 + lane;
    if (at >= hi) return;
    if (d.count_dev && at >= shard_size(d)) return;                                 // partition mode: `hi` is the host's bound, the list the owned agents
    const int agent = list[at];
    const Params P = agent_params(d, Pctx, agent);                                  // (a lane per agent: the agent's own time horizon / step / max speed)
    if (d.policy[agent] != POL_ORCA_LP) return;                                     // (partition mode hands over all owned agents)
    const PubRec me = d.rec[agent];
    if (me.flags & (FLAG_AT_GOAL | FLAG_COLLISION | FLAG_TIMEOUT)) return;          // mampenv.py:35 (k_solve wrote the bookkeeping)
    const Prep pr = ((const Prep *)d.prep)[agent];
    if (pr.bits & 1u) return;                                                       // bootstrap step: 0.3 v_pref, k_solve's branch
    const V3 pA = v3(me.px, me.py, me.pz);
    F3 vA; vA.x = me.vx; vA.y = me.vy; vA.z = me.vz;
    const V3 vpref = v3(pr.vpref[0], pr.vpref[1], pr.vpref[2]);
    const int K = d.nbr_valid[agent] ? d.nbr_n[agent] : 0;
    for (int j = 0; j < K; j++) {                                                   // orca3dPolicyOfficial.py:56-106
        const int nid = d.nbr_id[agent * K_MAX + j];
        V3 pB; F3 vB; double rB; const bool isob = (nid & NBR_OBSTACLE_BIT) != 0;
        if (isob) {
            const ObsRec o = d.obs[nid & ~NBR_OBSTACLE_BIT];
            pB = v3(o.px, o.py, o.pz); vB.x = vB.y = vB.z = 0.0f; rB = o.radius;
        } else {
            const PubRec o = d.rec[nid];
            pB = v3(o.px, o.py, o.pz); vB.x = o.vx; vB.y = o.vy; vB.z = o.vz; rB = o.radius;
        }
        const OrcaOb o = make_orca(P, pA, vA, me.radius, pB, vB, rB, isob);
        S.v[j][0][lane] = o.pl.p.x; S.v[j][1][lane] = o.pl.p.y; S.v[j][2][lane] = o.pl.p.z;
        S.v[j][3][lane] = o.pl.n.x; S.v[j][4][lane] = o.pl.n.y; S.v[j][5][lane] = o.pl.n.z;
    }
    LpAccess pl; pl.S = &S; pl.lane = lane;
    V3 nv = v3(0, 0, 0);
    const int pf = lp3(pl, K, P.max_speed, vpref, false, nv);                       // :108
    if (pf < K) {                                                                   // :110-111 linearProgram4: one wavefront, k_fallback
        const int slot = atomicAdd(d.fb_count, 1);
        d.fb_list[slot] = agent; d.is_fb[agent] = 1;
        return;
    }
    int32_t *diag = d.diag + (size_t)agent * 8;
    d.vpost[agent * 3 + 0] = nv.x; d.vpost[agent * 3 + 1] = nv.y; d.vpost[agent * 3 + 2] = nv.z;      // :113: not truncated
    diag[0] = -1; diag[1] = -1; diag[2] = -1; diag[3] = pf; diag[4] = 0;
    d.vpref_used[agent * 3 + 0] = vpref.x; d.vpref_used[agent * 3 + 1] = vpref.y; d.vpref_used[agent * 3 + 2] = vpref.z;
    const int st = ((pr.bits & 2u) ? ST_BAD_PREF_SPEED : 0);
    if (st) atomicOr(&d.status[agent], st);
}

__global__ __launch_bounds__(256) void k_prep(DeviceView d, Params P) {
    const int agent = blockIdx.x * blockDim.x + threadIdx.x;
    if (agent == 0) *d.fb_count = 0;                                       // start of a pass: empty fallback list
    if (agent < d.n) prep_agent<PREP_LIBM>(d, P, (Prep *)d.prep, agent);
}

// K2 epilogue, one LANE per agent: cartesian2spherical (util.py:44-55) -> float32 action row (mampenv.py:31,40), the
// is_collision flag of agent.py:84, and -- when the state stays resident -- update_velocitie (mampenv.py:83-105).
template <bool FUSE_INTEGRATE, bool INLINE_LIBM = false>
__device__ __forceinline__ void action_one(const DeviceView &d, const Params &P, int agent, bool derive) {
    PubRec me = d.rec[agent];
    float actf[7] = {0, 0, 0, 0, 0, 0, 0};
    if (!(me.flags & (FLAG_AT_GOAL | FLAG_COLLISION | FLAG_TIMEOUT))) {                // mampenv.py:35: else the row stays zero
        double act[7];
        V3 v;
        if (derive) {                              // k_solve left the index of the chosen candidate: scaPolicy.py:239 here, 64 agents per wavefront
            const Prep pr = ((const Prep *)d.prep)[agent];
            const int pol = d.policy[agent];
            CandTab T;
            T.num_N = (pol == POL_SCA && d.zaxis[agent]) ? 128 : 256;
            T.unit = (T.num_N == 256) ? d.unit256 : d.unit128;
            T.phi = nullptr;
            T.vp_idx = 2 * T.num_N;
            T.rad1 = pr.rad1;
            v = trunc5(cand_from_idx(T, d.diag[(size_t)agent * 8 + 2], v3(pr.vpref[0], pr.vpref[1], pr.vpref[2])));
            d.vpost[agent * 3] = v.x; d.vpost[agent * 3 + 1] = v.y; d.vpost[agent * 3 + 2] = v.z;
        } else v = v3(d.vpost[agent * 3], d.vpost[agent * 3 + 1], d.vpost[agent * 3 + 2]);
        cartesian2spherical<INLINE_LIBM>(d.heading[agent * 3 + 0], d.heading[agent * 3 + 1], v, d.policy[agent] == POL_ORCA_LP, act);
#pragma unroll
        for (int k = 0; k < 7; k++) actf[k] = (float)act[k];
        if (d.coll_new[agent]) { me.flags |= FLAG_COLLISION; d.rec[agent].flags = me.flags; }
        atomicAdd(&d.agent_steps[(agent & 255) * 16], 1ull);
    }
    float *out = d.action + (size_t)agent * 8;
#pragma unroll
    for (int k = 0; k < 7; k++) out[k] = actf[k];
    out[7] = 0.0f;
    if (d.trk_nbr0 && d.nbr_valid[agent]) d.trk_nbr0[agent] = d.nbr_n[agent] > 0 ? d.nbr_dsq[(size_t)agent * K_MAX] : -1.0;   // agent.py:79-99
    if (FUSE_INTEGRATE) integrate_agent<INLINE_LIBM>(d, P, agent, me, actf);
}

// K2 epilogue: one LANE per agent (see above), skipping the agents k_solve could not finish -- those are k_fallback's.
// (One launch with the fallback sweep until round 3: the sweep keeps all 513 candidates in registers, so the kernel carried 252
// VGPRs and the one-lane-per-agent epilogue ran at one wavefront per SIMD; alone it takes a quarter of that.)
template <bool FUSE_INTEGRATE>
__global__ __launch_bounds__(256) void k_action(DeviceView d, Params P) {
    SCA_TL(d, TL_ACTION);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= shard_size(d)) return;
    const int agent = shard_agent(d, idx);
    const int kind = d.is_fb[agent];
    if (kind == 1) return;
    action_one<FUSE_INTEGRATE, true>(d, P, agent, kind == 2);
}
// The agents without any suitable candidate (rare): one wavefront per entry of the fallback list runs the complete sweep (all
// 513 candidates in registers, incl. compute_without_suitV, scaPolicy.py:224-238) and then the same epilogue for that agent.
// The list's length is on the device: a fixed grid strides over it (an empty list costs an empty launch).  k_action and
// k_fallback never touch the same agent, so they need no order between them.
constexpr int FB_BLOCKS = 256;
// k_solve + k_fallback in ONE launch, for shards of so few agents that every wavefront is resident at once whatever its registers
// (round 4): an agent whose sweep leaves no suitable candidate is finished by its own wavefront on the spot instead of going
// through the list to a launch of its own -- which, list empty or not, sat between the solve and the epilogue on every pass's
// critical path (5 us + its gap of a 150-us step at N = 1024).  Passes with LP agents keep the list (k_lp's fallbacks need it).
__global__ __launch_bounds__(SOLVE_WAVES * 64) void k_solve_fb(DeviceView d, Params P) {
    SCA_TL(d, TL_SOLVE);
    __shared__ FastLds S;
    __shared__ SolveLds S2;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int idx = blockIdx.x * SOLVE_WAVES + wid;
    if (idx >= shard_size(d)) return;
    const int agent = shard_agent(d, idx);
    solve_fast<0, 1>(d, P, S, agent, lane, wid);
    __builtin_amdgcn_wave_barrier();
    int fb = 0;
    if (lane == 0) fb = d.is_fb[agent];                                  // (the lane that wrote it)
    if (__builtin_amdgcn_readfirstlane(fb) != 1) return;
    solve_one(d, P, S2, agent, lane, wid);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) d.is_fb[agent] = 0;                                   // the epilogue (k_action) takes it like any other agent
}
template <bool FUSE_INTEGRATE>
__global__ __launch_bounds__(SOLVE_WAVES * 64) void k_fallback(DeviceView d, Params P) {
    SCA_TL(d, TL_FALLBACK);
    __shared__ SolveLds S;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = *d.fb_count;
    for (int i = (int)blockIdx.x * SOLVE_WAVES + wid; i < n; i += FB_BLOCKS * SOLVE_WAVES) {
        const int agent = d.fb_list[i];
        solve_one(d, P, S, agent, lane, wid);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) action_one<FUSE_INTEGRATE>(d, P, agent, false);
        __builtin_amdgcn_wave_barrier();
    }
}

// k_fallback and k_action in ONE launch, for shards of so few agents that the epilogue's handful of workgroups does not care about its
// register allocation (round 6).  The two never touch the same agent and need no order between them (above); as two launches the
// fallback sweep -- 0.7 us of work at c3, its list empty in most passes -- cost a dispatch and its gap (6.5 us of a 70-us chain:
// profiles/r05_c3_auto_device_timeline.json) on every pass's critical path.  Workgroups [0, ablocks) are k_action's, the rest stride over
// the fallback list like k_fallback's.  Larger shards keep the two launches: there the one-lane-per-agent epilogue wants its own 66
// registers (k_action's note above), and the extra dispatch is a smaller share of the step.
constexpr int FB_BLOCKS_SMALL = 64;
template <bool FUSE_INTEGRATE>
__global__ __launch_bounds__(SOLVE_WAVES * 64) void k_action_fb(DeviceView d, Params P, int ablocks) {
    SCA_TL(d, TL_ACTION);
    __shared__ SolveLds S;
    if ((int)blockIdx.x < ablocks) {
        const int idx = blockIdx.x * blockDim.x + threadIdx.x;
        if (idx >= shard_size(d)) return;
        const int agent = shard_agent(d, idx);
        const int kind = d.is_fb[agent];
        if (kind == 1) return;
        action_one<FUSE_INTEGRATE, true>(d, P, agent, kind == 2);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = *d.fb_count;
    for (int i = ((int)blockIdx.x - ablocks) * SOLVE_WAVES + wid; i < n; i += FB_BLOCKS_SMALL * SOLVE_WAVES) {
        const int agent = d.fb_list[i];
        solve_one(d, P, S, agent, lane, wid);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) action_one<FUSE_INTEGRATE>(d, P, agent, false);
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(256) void k_integrate(DeviceView d, Params P) {
    SCA_TL(d, TL_ACTION);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= shard_size(d)) return;
    const int agent = shard_agent(d, idx);
    float act[7];
    for (int k = 0; k < 7; k++) act[k] = d.action[(size_t)agent * 8 + k];
    integrate_agent<true>(d, P, agent, d.rec[agent], act);
}

// check_agent_state (mampenv.py:61-80) + is_done (mampenv.py:51-59).  Candidates come from the kd-tree of the step's OLD
// positions: any pair that touches after the move was within r_a + r_b + 2 * max_step of each other before it.  K1 leaves
// the <= NEAR_MAX such objects of every agent it visits, so the usual case is EIGHT AGENTS PER WAVEFRONT, eight lanes each,
// one candidate per lane.  Agents without a list (K1 skipped them: bootstrap step; or the list overflowed) are traversed
// afterwards by the whole wavefront, one at a time.  The flags are committed into the moved record; the host swaps the
// two record buffers afterwards.
struct CollideCtx {
    PubRec me; V3 p, p_old; bool me_goal; int agent;
};
__device__ __forceinline__ bool collide_obstacle(const DeviceView &d, const CollideCtx &c, int o) {
    const ObsRec r = d.obs[o];
    return l3norm(c.p, v3(r.px, r.py, r.pz)) <= c.me.radius + r.radius;                    // mampenv.py:63-66
}
__device__ __forceinline__ bool collide_agent(const DeviceView &d, const CollideCtx &c, int j) {
    const PubRec rn = d.rec_new[j];
    const PubRec ro = d.rec[j];
    const double rs = c.me.radius + rn.radius;
    const V3 pn = v3(rn.px, rn.py, rn.pz);
    // new-new is seen by whichever of the two is checked second; the mixed pair by the first one
    bool hit = l3norm(c.p, pn) <= rs;
    if (j > c.agent) hit = hit || (l3norm(c.p, v3(ro.px, ro.py, ro.pz)) <= rs);            // agent moved, j not yet
    else hit = hit || (l3norm(pn, c.p_old) <= rs);                                         // j moved, agent not yet
    return hit && !c.me_goal;                                                              // mampenv.py:72-75
}
__device__ __forceinline__ CollideCtx collide_ctx(const DeviceView &d, int agent, PubRec &me_old) {
    CollideCtx c;
    c.agent = agent;
    c.me = d.rec_new[agent];
    me_old = d.rec[agent];
    c.p = v3(c.me.px, c.me.py, c.me.pz);
    c.p_old = v3(me_old.px, me_old.py, me_old.pz);
    c.me_goal = (c.me.flags & FLAG_AT_GOAL) != 0;
    return c;
}
// whole wavefront, one agent: range queries in both trees (wave-uniform agent)
__device__ __forceinline__ bool collide_traverse(const DeviceView &d, double agent_reach, double obs_reach, int *stack, int agent, int lane,
                                                 bool obstacles_only) {
    PubRec me_old;
    const CollideCtx c = collide_ctx(d, agent, me_old);
    bool hit = false;
    if (d.m > 0) {
        const double rq = c.me.radius + obs_reach;
        kd_traverse(d.owide, c.p, rq * rq, stack, lane, [&](int begin, int end) {
            if (lane < end - begin) hit = hit || collide_obstacle(d, c, d.operm[begin + lane]);
        });
    }
    if (obstacles_only) return __ballot(hit) != 0;                   // wave-uniform
    const double rq = c.me.radius + agent_reach;
    kd_traverse(d.awide, c.p_old, rq * rq, stack, lane, [&](int begin, int end) {
        if (lane < end - begin) { const int j = d.aperm[begin + lane]; if (j != agent) hit = hit || collide_agent(d, c, j); }
    });
    return __ballot(hit) != 0;
}

constexpr int K4_WAVES = 4;
constexpr int K4_APW = 64 / NEAR_MAX;      // agents per wavefront
static_assert(NEAR_MAX == 8, "k_collide_finish packs 8 lanes per agent");

// traverse(agent, obstacles_only) -> wave-uniform "touches something": the whole-wavefront fallback for one agent (kd-trees
// here, the grid in sca_grid.hip.h)
template <class Traverse>
__device__ __forceinline__ void collide_finish_body(const DeviceView &d, const Params &P, int check_arrived, Traverse traverse) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane & (NEAR_MAX - 1), grp = lane / NEAR_MAX;
    const int idx = (blockIdx.x * K4_WAVES + wid) * K4_APW + grp;
    const int cnt = shard_size(d);
    if (cnt <= 0) return;
    const bool exists = idx < cnt;
    const int agent = shard_agent(d, exists ? idx : cnt - 1);
    PubRec me_old;
    const CollideCtx c = collide_ctx(d, agent, me_old);
    const int near_n = d.near_n[agent];
    // an agent that had arrived or collided before this step did not move and cannot gain a flag: mampenv.py:72-75 never
    // flags an agent at its goal for touching another agent, a collided one is flagged already, and whether it touches an
    // obstacle (mampenv.py:63-66, checked for EVERY agent) was settled in the step it arrived -- unless the state came from
    // outside (sca_set_state with the at-goal flag set): then the first update after it looks once (check_arrived)
    const bool settled = (me_old.flags & (FLAG_AT_GOAL | FLAG_COLLISION)) != 0;
    const bool arrived_only = check_arrived && d.m > 0 && (me_old.flags & (FLAG_AT_GOAL | FLAG_COLLISION)) == FLAG_AT_GOAL;
    bool hit = false;
    if (exists && !settled && sub < near_n) {
        const int id = d.near_id[(size_t)agent * NEAR_MAX + sub];
        hit = (id & NBR_OBSTACLE_BIT) ? collide_obstacle(d, c, id & ~NBR_OBSTACLE_BIT) : collide_agent(d, c, id);
    }
    bool any = ((__ballot(hit) >> (grp * NEAR_MAX)) & ((1ull << NEAR_MAX) - 1ull)) != 0;
    unsigned long long todo = __ballot(exists && sub == 0 && ((!settled && near_n < 0) || arrived_only));
    while (todo) {
        const int l0 = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int ag = __builtin_amdgcn_readlane(agent, l0);
        const bool obs_only = __builtin_amdgcn_readlane((int)arrived_only, l0) != 0;
        const bool r = traverse(ag, obs_only);
        if (lane / NEAR_MAX == l0 / NEAR_MAX) any = r;
    }
    if (exists && sub == 0) {
        uint32_t f = c.me.flags;
        if (any) f |= FLAG_COLLISION;
        if (d.total_dist[agent] > d.max_run_dist[agent]) f |= FLAG_TIMEOUT;               // mampenv.py:77-79
        const V3 g = v3(d.goal[agent * 3], d.goal[agent * 3 + 1], d.goal[agent * 3 + 2]);
        if (l3norm(c.p, g) <= P.near_goal_threshold) f |= FLAG_AT_GOAL;                   // mampenv.py:53-54
        d.rec_new[agent].flags = f;
        if (!(f & (FLAG_AT_GOAL | FLAG_COLLISION | FLAG_TIMEOUT))) atomicAdd(&d.done_count[(agent & 255) * 32], 1);
    }
}

__global__ __launch_bounds__(K4_WAVES * 64) void k_collide_finish(DeviceView d, Params P, double agent_reach, double obs_reach,
                                                                int check_arrived) {
    SCA_TL(d, TL_COLLIDE);
    __shared__ int stacks[K4_WAVES][KD_STACK];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    collide_finish_body(d, P, check_arrived, [&](int ag, bool obs_only) {
        return collide_traverse(d, agent_reach, obs_reach, stacks[wid], ag, lane, obs_only);
    });
}

// the near lists belong to the policy pass of the same step; without one, k_collide_finish must traverse
__global__ __launch_bounds__(256) void k_invalidate_near(DeviceView d) {
    const int agent = blockIdx.x * blockDim.x + threadIdx.x;
    if (agent < d.n) d.near_n[agent] = -1;
}

// agent.neighbors[0][1] of the last pass (what the v_pref tracker reads, scaPolicy.py:299): -1 = empty list, -2 = the pass
// did not touch the list (agent skipped: keep the previous value)
__global__ __launch_bounds__(256) void k_nbr0(DeviceView d, double *out) {
    const int agent = blockIdx.x * blockDim.x + threadIdx.x;
    if (agent >= d.n) return;
    out[agent] = d.nbr_valid[agent] ? (d.nbr_n[agent] > 0 ? d.nbr_dsq[(size_t)agent * K_MAX] : -1.0) : -2.0;
}

// multi-GPU only: agents of other shards arrived by all-gather with the flags their owner published one step ago;
// replicate the at-goal test for them -- the only flag of another agent the policy reads (scaPolicy.py:53).
__global__ __launch_bounds__(256) void k_goal_flags_others(DeviceView d, Params P) {
    SCA_TL(d, TL_GOAL_FLAGS);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int agent;
    if (d.present) {                                                     // partition mode: the halo copies, behind the owned agents
        const int own = shard_size(d);
        if (own + i >= present_count(d)) return;
        agent = d.present[own + i];
    } else {
        agent = i;
        if (agent >= d.n || shard_owns(d, agent)) return;
    }
    const PubRec r = d.rec_new[agent];
    const V3 g = v3(d.goal[agent * 3], d.goal[agent * 3 + 1], d.goal[agent * 3 + 2]);
    if (l3norm(v3(r.px, r.py, r.pz), g) <= P.near_goal_threshold) d.rec_new[agent].flags = r.flags | FLAG_AT_GOAL;
}

}  // namespace sca
