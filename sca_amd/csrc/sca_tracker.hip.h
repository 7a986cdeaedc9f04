// sca_tracker.hip.h -- SCA's preferred-velocity tracker on the device (second stage).
//
// compute_v_pref of SCAPolicy / RVO3dDubinsPolicy (mamp/policies/sca/scaPolicy.py:264-338) runs first in find_next_action
// (:32) and depends on the agent's own state only (position, velocity, heading, its tracker record and the distSq of the
// first neighbour the PREVIOUS pass left, :299), so all agents of a step are independent: one lane per agent.
//
//   k_track   every active SCA / RVO3D+Dubins agent of the shard: update_dubins, the follow-or-re-plan decision and, for the
//             followers, v_pref.  Agents that must re-plan are appended to a list (wave-aggregated atomic).
//   k_replan  one lane per listed agent: the 3-D Dubins planner (dubinsmaneuver3d.py:34-162 -- a scalar search of ~50-100
//             2-D plans, ~10^5 fp64 instructions), the first tracked node, v_pref.  Compacting the re-planning agents into
//             dense wavefronts keeps the long plans from stalling the followers that would share their wave.
//   k_replan_few  the same with 4, 16, 32 or 64 lanes per plan, for passes with few re-plans (see below).
//   k_replan_mid  four lanes per plan at two wavefronts per SIMD, for the counts in between.
//   k_track_replan  k_track + k_replan in one launch (no list) for passes in which nearly the whole shard re-plans.
// They run on the library's main stream -- they are the pass's critical path; the kd build, the neighbour query and the
// v_pref-independent half of the solve run beside them on a second stream, joined before the prologue that reads v_pref.
// The host launches the re-plan kernels a recent pass's count makes possible (launch_tracker); the device-side count of the
// pass decides which of them, and which form inside k_replan_few, does the work.
//
// The arithmetic is sca_dubins.hpp compiled for gfx950: same statements as the host tracker and the same libm -- glibc 2.35's
// sin / cos / atan2 / acos / pow restated operation for operation (sca_glibc_math.h), so every plan and every v_pref equals the
// host tracker's bit for bit.  State: one AgentTrack record per agent, resident in HBM.
#pragma once
#include <hip/hip_runtime.h>
#include "sca_dubins.hpp"
#include "sca_kernels.hip.h"
#include "sca_spec_trees.h"

namespace sca {

struct TrackDev {
    sca_dubins::AgentTrack *st;   // [n]
    double *nbr0;                 // [n] distSq of agent.neighbors[0] as the previous pass left it (-1: empty list)
    int32_t *list;                // [TRK_BUCKETS][n] agents that re-plan in this pass, binned by how many candidate radii their PREVIOUS search
                                  // tried (bucket b holds its members at list[b * n ..]): the lane-per-plan kernel takes them longest first
    int32_t *bcount;              // [4][TRK_BUCKETS] members per bucket, the same ring over the passes as `count`
    int n;                        // agents (the stride of the buckets)
    int32_t *count;               // [4] re-plans of the pass (= list length), a ring over the passes: pass p counts in slot p & 3 and
                                  // zeroes slot (p + 1) & 3; the host reads the previous pass's slot, which is final, without waiting
    int parity;                   // the slot of this pass
    int nbr0_from_lists;          // 1: take nbr0 from the neighbour lists as they are (sca_device_tracker_vpref without an upload); inside a
                                  // pass it is what the previous pass's epilogue saved (DeviceView::trk_nbr0)
    int lo, hi;                   // this launch takes the pass when lo < (re-plans of the pass) <= hi; the launches of a pass cover every count
    int prep;                     // 1: the kernels also write the solve's per-agent prologue (inside a pass whose neighbour branch overlaps)
    Params P;
    int mid_max;                  // TRK_MID_MAX unless overridden (SCA_TRK_MID_MAX, tuning): the quad form's upper end
    int spec2_max, spec3_max, spec4_max;   // TRK_SPEC*_MAX unless overridden (SCA_TRK_SPEC2_MAX ..., tuning)
};

// Which kernel re-plans a pass, by the pass's re-plan count (read on the device; the host launches the kernels whose range a recent
// count makes possible and widens the outermost ranges so that every count is covered):
//   <= TRK_SPEC4_MAX / SPEC3 / SPEC2   k_replan_group<64 / 32 / 16>: the search 4 / 3 / 2 steps per round (below)
//   <= TRK_MID_MAX    k_replan_group<4>: four lanes per plan, two wavefronts per SIMD (217 registers): 32 768 plans = 2048 wavefronts
//   above             k_replan / k_track_replan: one lane per plan (0.44 ms up to 65 536 plans, 0.63 up to 131 072)
// Re-plans ordered by expected length.  A search is sequential and takes 48 ... 114 candidate radii; a wavefront lasts as long as
// its longest lane, and ~96 000 plans are 1500 wavefronts on 1024 SIMDs -- the SIMDs that hold two wavefronts run each at little
// more than half speed, and the kernel lasts as long as the slowest of them.  With the plans in descending order of the number of
// candidates their previous search took (the same agent one step later: correlation 0.9), the long searches are dispatched
// first and sit alone on their SIMDs, the wavefronts that double up are the short ones, and the lanes of a wavefront end
// together.  Measured on the planner alone (tools/bench/plan_bench.hip, 96 256 plans): 1.12 ms in arbitrary order, 0.76 ms in
// descending order.  Buckets of eight candidates: bucket = clamp((iters - 40) / 8, 0, TRK_BUCKETS - 1), filled by k_track.
constexpr int TRK_BUCKETS = 12;
__device__ __forceinline__ int trk_bucket(int prev_iters) { const int b = (prev_iters - 40) >> 3; return b < 0 ? 0 : (b >= TRK_BUCKETS ? TRK_BUCKETS - 1 : b); }
// the idx-th re-planning agent of the pass, longest expected search first
__device__ __forceinline__ int trk_list_agent(const int32_t *list, const int32_t *bc, int n, int idx) {
    int rem = idx;
#pragma unroll
    for (int b = TRK_BUCKETS - 1; b > 0; b--) {
        const int c = bc[b];
        if (rem < c) return list[(size_t)b * n + rem];
        rem -= c;
    }
    return list[rem];
}
constexpr int TRK_MID_MAX = 32768;
constexpr int TRK_REPLAN_LANES = 256;     // four wavefronts per workgroup = one per SIMD of a CU: the dispatcher then loads the SIMDs evenly
                                          // (65 536 plans as 1024 one-wave workgroups: 0.63 ms, some SIMDs drew two; as 256 of these: 0.44)
                                          // and, above one wave per SIMD, doubles up whole CUs, which leaves the others room for the
                                          // 512-thread workgroups of the kd build running beside the re-plans (k_kd_block 154 -> 99 us at c4)

__device__ __forceinline__ bool track_active(const DeviceView &d, int agent) { return tracker_owns(d, agent); }   // mampenv.py:35
__device__ __forceinline__ void track_store(const DeviceView &d, const TrackDev &K, int agent, const double *V) {
    for (int q = 0; q < 3; q++) {
        double x = V[q];
        if (x != x) x = 0.0;                                             // what numpy.nan_to_num does on the host path
        else if (x > 1.7976931348623157e308) x = 1.7976931348623157e308;
        else if (x < -1.7976931348623157e308) x = -1.7976931348623157e308;
        d.vpref_ext[agent * 3 + q] = x;
    }
    // inside a policy pass: the agent's prologue for the solve (it reads the v_pref just stored), by the lane that has it --
    // a launch of its own behind the join (k_prep_shard, 6.5 us + its gap on the critical path) until the end of round 2
#ifdef SCA_TRK_PREP_CALL                                                    // (A/B: the arctangent as a call into the constant-table copy)
    if (K.prep) prep_agent<0>(d, K.P, (Prep *)d.prep, agent);
#else
    if (K.prep) prep_agent<1>(d, K.P, (Prep *)d.prep, agent);                 // (every kernel that stores a v_pref has the libm tables in LDS)
#endif
}

#ifdef SCA_KT_TIMING    // per-phase wall-clock ticks of workgroup 0's thread 0 into d.kdq_list (debug builds only)
#define KT_MARK() do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long t_ = wall_clock64(); d.kdq_list[kt_i++] = (int)(t_ - kt_t); kt_t = t_; } } while (0)
#define KT_MARK_INIT() int kt_i = 0; long long kt_t = wall_clock64()
#else
#define KT_MARK() do { } while (0)
#define KT_MARK_INIT() do { } while (0)
#endif
__global__ __launch_bounds__(256) void k_track(DeviceView d, sca_dubins::TrackView T, TrackDev K) {
    SCA_TL(d, TL_TRACK);
    // the re-plan list: per workgroup the lanes count themselves into their buckets in LDS, ONE vector atomic fetches the
    // workgroup's offsets in all twelve buckets, and the lanes write their slots.  (Until the end of round 3 every wavefront
    // fetched its offsets itself, one bucket after the other: ~5000 dependent same-address atomics per pass, which is what
    // k_track's 52 us at c4 were made of -- it executes 1400 instructions per wavefront.)
    __shared__ int s_cnt[TRK_BUCKETS], s_base[TRK_BUCKETS];
    KT_MARK_INIT();
    sca_gm::lds_tables_load();                                           // atan2's and sin / cos's tables into LDS (all threads, first)
    KT_MARK();
    if (blockIdx.x == 0 && threadIdx.x == 0) K.count[(K.parity + 1) & 3] = 0;
    if (blockIdx.x == 0 && threadIdx.x < TRK_BUCKETS) K.bcount[((K.parity + 1) & 3) * TRK_BUCKETS + threadIdx.x] = 0;
    if (threadIdx.x < TRK_BUCKETS) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    KT_MARK();
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_shard = idx < shard_size(d);
    const int agent = in_shard ? shard_agent(d, idx) : 0;
    bool active = false, replan = false;
    int mine = 0, my_rank = 0;
    double pos[3] = {0, 0, 0}, dif[3] = {0, 0, 0};
    if (in_shard) {
        double nb0 = K.nbr0[agent];
        if (K.nbr0_from_lists && d.nbr_valid[agent]) {                   // lists of the previous pass (agent.py:79-99)
            nb0 = d.nbr_n[agent] > 0 ? d.nbr_dsq[(size_t)agent * K_MAX] : -1.0;
            K.nbr0[agent] = nb0;
        }
        active = track_active(d, agent);
        if (active) {
            const PubRec r = d.rec[agent];
            pos[0] = r.px; pos[1] = r.py; pos[2] = r.pz;
            const float vel[3] = {r.vx, r.vy, r.vz};
            sca_dubins::AgentTrack &a = K.st[agent];
            KT_MARK();
            replan = sca_dubins::track_decide(T, a, agent, pos, vel, nb0, dif);
            KT_MARK();
            if (replan) {
                mine = trk_bucket(a.plan.iters);                         // the bucket of the agent's previous search (a.plan still holds it)
                my_rank = atomicAdd(&s_cnt[mine], 1);
            } else {
                double V[3];
                sca_dubins::track_finish(T, a, agent, pos, dif, V);
                track_store(d, K, agent, V);
            }
            KT_MARK();
        }
    }
    __syncthreads();
    KT_MARK();
    if (threadIdx.x < TRK_BUCKETS) {
        const int c = s_cnt[threadIdx.x];
        s_base[threadIdx.x] = c > 0 ? atomicAdd(&K.bcount[K.parity * TRK_BUCKETS + threadIdx.x], c) : 0;
    } else if (threadIdx.x == 64) {
        int total = 0;
        for (int b = 0; b < TRK_BUCKETS; b++) total += s_cnt[b];
        if (total > 0) atomicAdd(&K.count[K.parity], total);
    }
    __syncthreads();
    if (replan) K.list[(size_t)mine * K.n + s_base[mine] + my_rank] = agent;
    KT_MARK();
}

// two wavefronts per SIMD, 256 registers each.  Measured (round 5, -DSCA_REPLAN_WAVES=3 / 4): 168 registers + 119 spilled / 128 + 291 spilled,
// c4 0.672 -> 0.88 / 1.36 ms per step: a candidate's working set is the registers, and scratch traffic costs more than a third wavefront hides.
#ifndef SCA_REPLAN_WAVES
#define SCA_REPLAN_WAVES 2
#endif
__global__ __launch_bounds__(TRK_REPLAN_LANES, SCA_REPLAN_WAVES) void k_replan(DeviceView d, sca_dubins::TrackView T, TrackDev K) {
    SCA_TL(d, TL_REPLAN);
    sca_gm::lds_tables_load();                                           // atan2's and sin / cos's tables into LDS (all threads, first)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int count = K.count[K.parity];
    if (idx >= count || count <= K.lo) return;                              // fewer re-plans: k_replan_few's / k_replan_mid's pass
    const int agent = trk_list_agent(K.list, K.bcount + K.parity * TRK_BUCKETS, K.n, idx);
    if (T.cls && T.cls[agent] != T.class_id) return;                       // another class's launch plans this agent (TrackView)
    const PubRec r = d.rec[agent];
    const double pos[3] = {r.px, r.py, r.pz};
    const double heading[3] = {d.heading[agent * 3], d.heading[agent * 3 + 1], d.heading[agent * 3 + 2]};
    sca_dubins::AgentTrack &a = K.st[agent];
    double dif[3], V[3];
    sca_dubins::track_replan(T, a, agent, pos, heading, dif);
    sca_dubins::track_finish(T, a, agent, pos, dif, V);
    track_store(d, K, agent, V);
}

// k_track + k_replan in ONE launch, one lane per agent of the shard: for passes in which nearly every tracked agent re-plans
// (the circle: 96 %), where compacting the re-planners into a list buys nothing and costs a launch on the pass's critical
// path.  Lanes that follow their path finish early inside their wavefront.  Only counts the re-plans (no list).
__global__ __launch_bounds__(TRK_REPLAN_LANES, 2) void k_track_replan(DeviceView d, sca_dubins::TrackView T, TrackDev K) {
    SCA_TL(d, TL_REPLAN);
    sca_gm::lds_tables_load();                                           // atan2's and sin / cos's tables into LDS (all threads, first)
    if (blockIdx.x == 0 && threadIdx.x == 0) K.count[(K.parity + 1) & 3] = 0;
    // (the next pass may be k_track's again: it appends to the bucket counts of that parity)
    if (blockIdx.x == 0 && threadIdx.x < TRK_BUCKETS) K.bcount[((K.parity + 1) & 3) * TRK_BUCKETS + threadIdx.x] = 0;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= shard_size(d)) return;
    const int agent = shard_agent(d, idx);
    const double nb0 = K.nbr0[agent];                                    // saved by the previous pass's epilogue
    if (!track_active(d, agent)) return;
    if (T.cls && T.cls[agent] != T.class_id) return;                     // (decision AND plan by the agent's own class's launch)
    const PubRec r = d.rec[agent];
    const double pos[3] = {r.px, r.py, r.pz};
    const float vel[3] = {r.vx, r.vy, r.vz};
    sca_dubins::AgentTrack &a = K.st[agent];
    double dif[3], V[3];
    const bool replan = sca_dubins::track_decide(T, a, agent, pos, vel, nb0, dif);
    const unsigned long long m = __ballot(replan);
    if (m != 0 && (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(&K.count[K.parity], __popcll(m));
    if (replan) {
        const double heading[3] = {d.heading[agent * 3], d.heading[agent * 3 + 1], d.heading[agent * 3 + 2]};
        sca_dubins::track_replan(T, a, agent, pos, heading, dif);
    }
    sca_dubins::track_finish(T, a, agent, pos, dif, V);
    track_store(d, K, agent, V);
}

// ---- four lanes per plan ---------------------------------------------------------------------------------------------------
// One re-plan is a strictly sequential search (every candidate radius depends on the verdict on the previous one), ~65 and up
// to ~115 candidates of two 2-D plans each; a lane working alone needs 0.6-0.8 ms for it, whatever the size of the swarm.
// Inside one candidate, though, the four CSC words of a 2-D plan are independent, and so are the three sin/cos pairs of its
// frame: a DPP quad evaluates them side by side (csc_word_uniform: one instruction stream, per-lane signs), the winner is
// picked with the reference's first-minimum rule from quad broadcasts, and everything else is computed redundantly by the
// four lanes, which therefore never diverge.  ~3x shorter critical path; 4x the lanes, so ~1.3x the total work: used while
// the re-plans of a pass leave SIMDs idle anyway (count <= TRK_QUAD_MAX), the lane-per-plan kernel takes over above that.

template <int K> __device__ __forceinline__ double quad_bcast_d(double x) {
    constexpr int CTRL = K | (K << 2) | (K << 4) | (K << 6);               // quad_perm:[K,K,K,K]
    const unsigned long long v = (unsigned long long)__double_as_longlong(x);
    const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)v, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)(v >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double lane_fetch_d(double x, int src_lane) {
    const unsigned long long v = (unsigned long long)__double_as_longlong(x);
    const int lo = __shfl((int)(unsigned)v, src_lane), hi = __shfl((int)(unsigned)(v >> 32), src_lane);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

struct QuadFrame { sca_dubins::Frame2D F; double mbeta; };

// frame2d with the three sincos arguments (alpha, beta, alpha - beta) on lanes 0, 1, 2 of the quad
__device__ __forceinline__ QuadFrame frame2d_quad(const double start[3], const double end[3], int sub) {
    using namespace sca_dubins;
    QuadFrame Q;
    const double ex = end[0] - start[0], ey = end[1] - start[1];
    Q.F.D = ::sqrt(m_pow(ex, 2.0) + m_pow(ey, 2.0));
    const double theta = mod2pi(m_atan2(ey, ex));
    Q.F.alpha = mod2pi(start[2] - theta);
    Q.F.beta = mod2pi(end[2] - theta);
    const double arg = sub == 0 ? Q.F.alpha : (sub == 1 ? Q.F.beta : Q.F.alpha - Q.F.beta);
    double sn, cs;
    m_sincos(arg, sn, cs);
    Q.F.sa = quad_bcast_d<0>(sn);
    Q.F.ca = quad_bcast_d<0>(cs);
    Q.F.sb = quad_bcast_d<1>(sn);
    Q.F.cb = quad_bcast_d<1>(cs);
    Q.F.c_ab = quad_bcast_d<2>(cs);
    Q.mbeta = mod2pi(Q.F.beta);
    return Q;
}

// plan2d: lane `sub` of the quad evaluates CSC word `sub`; RLR / LRL (feasible only for end points closer than four radii) by
// all four lanes alike
__device__ __forceinline__ sca_dubins::Maneuver2D plan2d_quad(const QuadFrame &Q, double yaw, double c, int sub, int lane) {
    using namespace sca_dubins;
    const Frame2D &F = Q.F;
    Maneuver2D m;
    m.yaw = yaw; m.r_min = c; m.t = m.p = -1.0; m.length = INFINITY; m.ok = false;
    m.mode[0] = m.mode[1] = m.mode[2] = 0;
    const double d = F.D / c;
    double t, p, q;
    const bool ok = csc_word_uniform(sub, F.alpha, F.beta, Q.mbeta, d, F.sa, F.sb, F.ca, F.cb, F.c_ab, t, p, q);
    const double cost = ok ? c * (::fabs(t) + ::fabs(p) + ::fabs(q)) : INFINITY;     // an infeasible word is skipped (:200-203)
    const double c0 = quad_bcast_d<0>(cost);
    const double c1 = quad_bcast_d<1>(cost);
    const double c2 = quad_bcast_d<2>(cost);
    const double c3 = quad_bcast_d<3>(cost);
    int w = -1;
    double bcost = INFINITY;                                                         // `if bcost > cost` in planner order
    if (bcost > c0) { w = 0; bcost = c0; }
    if (bcost > c1) { w = 1; bcost = c1; }
    if (bcost > c2) { w = 2; bcost = c2; }
    if (bcost > c3) { w = 3; bcost = c3; }
    const int src = (lane & ~3) | (w & 3);
    double bt = lane_fetch_d(t, src);
    double bp = lane_fetch_d(p, src);
    for (int k = 4; k < 6; k++) {
        double t2, p2, q2; char md[3];
        if (!word(k, F.alpha, F.beta, d, F.sa, F.sb, F.ca, F.cb, F.c_ab, t2, p2, q2, md)) continue;
        const double cost2 = c * (::fabs(t2) + ::fabs(p2) + ::fabs(q2));
        if (bcost > cost2) { w = k; bcost = cost2; bt = t2; bp = p2; }
    }
    if (w >= 0) {
        m.ok = true; m.t = bt; m.p = bp;
        m.mode[0] = (w == 0 || w == 2 || w == 5) ? 'L' : 'R';
        m.mode[1] = w < 4 ? 'S' : (w == 4 ? 'L' : 'R');
        m.mode[2] = (w == 0 || w == 3 || w == 5) ? 'L' : 'R';
    }
    m.length = bcost;
    return m;
}

// frame2d_quad for the vertical plane: start (0, z_i, pitch_i), end (len, z_f, pitch_f); dz^2 = pow(z_f - z_i, 2) is the same for
// every candidate of a search and comes from SearchConst (same argument, same function, same bits: sca_dubins::frame2d_vertical)
__device__ __forceinline__ QuadFrame frame2d_quad_vertical(double len, double dz, double dz2, double spitch, double epitch, int sub) {
    using namespace sca_dubins;
    QuadFrame Q;
    const double ex = len - 0.0;
    Q.F.D = ::sqrt(m_pow(ex, 2.0) + dz2);
    const double theta = mod2pi(m_atan2(dz, ex));
    Q.F.alpha = mod2pi(spitch - theta);
    Q.F.beta = mod2pi(epitch - theta);
    const double arg = sub == 0 ? Q.F.alpha : (sub == 1 ? Q.F.beta : Q.F.alpha - Q.F.beta);
    double sn, cs;
    m_sincos(arg, sn, cs);
    Q.F.sa = quad_bcast_d<0>(sn);
    Q.F.ca = quad_bcast_d<0>(cs);
    Q.F.sb = quad_bcast_d<1>(sn);
    Q.F.cb = quad_bcast_d<1>(cs);
    Q.F.c_ab = quad_bcast_d<2>(cs);
    Q.mbeta = mod2pi(Q.F.beta);
    return Q;
}
__device__ __forceinline__ int try_to_construct_quad(const QuadFrame &H, const sca_dubins::SearchConst &K, const double qi[5], const double qf[5], double Rmin,
                                                     const double pitchlims[2], double hr, sca_dubins::Maneuver2D &mh,
                                                     sca_dubins::Maneuver2D &mv, int sub, int lane) {
    using namespace sca_dubins;
    (void)Rmin;
    const double vc = ::sqrt(K.inv_rmin2 - 1.0 / m_pow(hr, 2.0));          // (1 / Rmin^2 once per search: SearchConst)
    if (vc < 1e-5) return 0;                                             // see try_to_construct: mh would not be read
    mh = plan2d_quad(H, qi[3], hr, sub, lane);
    const double vr = 1.0 / vc;
    const QuadFrame V = frame2d_quad_vertical(mh.length, qf[2] - qi[2], K.dz2, qi[4], qf[4], sub);
    mv = plan2d_quad(V, qi[4], vr, sub, lane);
    if (mv.mode[0] == 'R' && mv.mode[1] == 'L' && mv.mode[2] == 'R') return 0;
    if (mv.mode[0] == 'R') { if (qi[4] - mv.t < pitchlims[0]) return 0; }
    else { if (qi[4] + mv.t > pitchlims[1]) return 0; }
    return 2;
}

// ---- the quad search, lean (round 3) ------------------------------------------------------------------------------------------
// What sca_dubins::lean::candidate is to the lane-per-plan search, for a quad: a candidate is evaluated for what the search reads
// of it (feasible?, length); when every lane of the wavefront is far (d >= 7 in both planes) lane `sub` evaluates word `sub`
// with the lean pieces -- case (i) arctangents in rows (the word's own and, for LSR / RSL, the one of 2 / p; LSL / RSR run
// atan2(+0, p) = +0 through the same instructions), sqrt without its special cases, one sin / cos pair per lane for the
// vertical frame -- and the costs meet through DPP broadcasts.  Anything else goes through try_to_construct_quad.  The
// winner's maneuvers are constructed once at the end of the search (try_to_construct_quad on the accepted radius).
struct QuadWordSigns { bool cross, rfirst, negv, negya, negyb, negq, usemb; double y2; };
__device__ __forceinline__ QuadWordSigns quad_word_signs(int sub) {
    QuadWordSigns g;
    g.cross = sub >= 2; g.rfirst = (sub & 1) != 0; g.negv = sub == 0 || sub == 3; g.negya = sub == 0 || sub == 2;
    g.negyb = sub == 1 || sub == 2; g.negq = g.negyb; g.usemb = sub == 2;
    g.y2 = sub == 2 ? -2.0 : (sub == 3 ? 2.0 : 0.0);                     // LSL / RSR: atan2(+0, p)
    return g;
}
// word `sub` of a far 2-D problem: cost, first segment; kmin as in lean::atan_far_n
__device__ __forceinline__ double quad_word_far(const sca_dubins::Frame2D &F, double mbeta, double d, double c, const QuadWordSigns &g,
                                                uint32_t &kmin, double &t_out) {
    using namespace sca_dubins;
    using sca_gm::flip;
    const double cab2 = 2 * F.c_ab, d2 = d * d, dd = 2 * d;
    const double u = flip(F.sa, g.rfirst), v = flip(F.sb, g.negv);
    const double S = u + v;
    const double k2 = g.cross ? -2.0 : 2.0;
    const double p2[1] = {((k2 + d2) + flip(cab2, !g.cross)) + (dd * S)};
    double p[1];
    lean::sqrt_pos_n<1>(p2, p);
    const double y[2] = {flip(F.ca, g.negya) + flip(F.cb, g.negyb), g.y2}, x[2] = {(d + u) + v, p[0]};
    double A[2];
    lean::atan_far_n<2, 1>(y, x, A, kmin);
    const double tmp = A[0] - A[1];                                       // (A[1] = +0 for LSL / RSR: A[0] - (+0) == A[0])
    const double ta = tmp - F.alpha, qa = (g.usemb ? mbeta : F.beta) - tmp;
    const double arg[2] = {flip(ta, g.rfirst), flip(qa, g.negq)};
    double m[2];
    lean::mod2pi_n<2>(arg, m);
    t_out = m[0];
    return c * (::fabs(m[0]) + ::fabs(p[0]) + ::fabs(m[1]));
}
// returns feasible?; len = the path's length.  All four lanes of the quad return the same.
// Round 4: the lean pieces are taken PER PLANE.  Round 3 took the lean form only for candidates that were far in both planes and
// otherwise evaluated the whole candidate the literal way -- after having evaluated it the lean way first: c2 / c5 search their last
// ~40 candidates at d_V = D_V / vr < 7 (vr grows without bound as the radius approaches Rmin) and ran 12 % / 20 % SLOWER than a
// build without any lean form.  Now:
//   (i)   a bound first: the horizontal path is at most D_H + (4 pi + 3) hr long (two arcs of less than a turn, a straight of at most
//         d + 3 radii), hence d_V <= sqrt((D_H + 16 hr)^2 + dz^2) * vc; below 7 for some lane: no attempt at the far vertical block;
//   (ii)  d_V itself as soon as the horizontal length is known, before the vertical frame's arctangent, sin / cos and words;
//   (iii) far in the horizontal plane but near in the vertical one (the end of every search on paths of a few hundred metres: c2): the
//         lean horizontal length IS plan2d's (same words, same first-minimum), so only the vertical maneuver goes the literal way --
//         frame2d_quad_vertical + plan2d_quad on that length, then try_to_construct's three tests (dubinsmaneuver3d.py:152-161).
//         The vertical FRAME is lean in (iii) as well (it does not depend on d_V): k_replan_group<64> 156 -> 146 us at c2.  The
//         64-lane kernel sits at 256 registers with it (one build of this form needed 261, i.e. one wavefront per SIMD: c5 108 ->
//         129 us), which is why TRK_SPEC4_MAX is 1024 now: a wavefront per plan only while every plan gets a SIMD of its own; 1025 ..
//         1280 plans take the 32-lane form (measured equal at c5, whose count sits there).  The 32- / 16- / 4-lane kernels: 245 / 233 / 230.
// All of it only chooses between evaluations that return the same bits.
__device__ __forceinline__ bool cand_quad(bool fast_ok, const QuadFrame &H, const sca_dubins::SearchConst &K, const double qi[5], const double qf[5],
                                          double Rmin, const double pitchlims[2], double hr, int sub, int lane, const QuadWordSigns &g, double &len) {
    using namespace sca_dubins;
    if (fast_ok && !sca_dubins::lean::any_says(hr != Rmin)) { len = 0.0; return false; }      // (see lean::candidate: the radius Rmin itself)
    const double dH = H.F.D / hr;
    if (fast_ok && !sca_dubins::lean::any_says(!lean::far_d(dH))) {
        const double vc = sca_gm::sqrt_(K.inv_rmin2 - 1.0 / sca_gm::g_pow2_main(hr));
        const bool flat = vc < 1e-5;
        const double dz = qf[2] - qi[2];
        const double reach = H.F.D + 16.0 * hr;
        const double dV_ub = ::sqrt(reach * reach + dz * dz) * vc;
        const bool try_far_v = !sca_dubins::lean::any_says(!flat && dV_ub < 7.0);
        uint32_t kmin = 0xffffffffu;
        double tw;
        const double costH = quad_word_far(H.F, H.mbeta, dH, hr, g, kmin, tw);
        const double lenH = sca_gm::min_(sca_gm::min_(quad_bcast_d<0>(costH), quad_bcast_d<1>(costH)), sca_gm::min_(quad_bcast_d<2>(costH), quad_bcast_d<3>(costH)));
        const double vr = 1.0 / (flat ? 1.0 : vc);
        Frame2D F;
        F.D = lean::sqrt_pos(sca_gm::g_pow2_main(lenH) + K.dz2);
        const bool far_theta = lenH > ::fabs(dz) && lenH < 1.2676506002282294e30;
        // the vertical frame by the lean pieces whenever the path is longer than it is high (theta = atan2(dz, lenH) is then a case-(i)
        // arctangent): it does not care how near the end points are in units of the vertical radius -- only the words do
        if (!sca_dubins::lean::any_says(!flat && !far_theta)) {
            const double y1[1] = {dz}, x1[1] = {lenH};
            double th[1];
            lean::atan_far_n<1, 1>(y1, x1, th, kmin);
            const double theta = mod2pi(th[0]);
            F.alpha = mod2pi(qi[4] - theta);
            F.beta = mod2pi(qf[4] - theta);
            const double a1[1] = {sub == 0 ? F.alpha : (sub == 1 ? F.beta : F.alpha - F.beta)};
            double s1[1], c1[1];
            lean::sincos_n<1>(a1, s1, c1);
            F.sa = quad_bcast_d<0>(s1[0]); F.ca = quad_bcast_d<0>(c1[0]);
            F.sb = quad_bcast_d<1>(s1[0]); F.cb = quad_bcast_d<1>(c1[0]);
            F.c_ab = quad_bcast_d<2>(c1[0]);
            const double dV = F.D / vr;
            const double mbetaV = mod2pi(F.beta);
            if (try_far_v && !sca_dubins::lean::any_says(!flat && !lean::far_d(dV))) {
                const double costV = quad_word_far(F, mbetaV, dV, vr, g, kmin, tw);
                // the reference's first-minimum rule over the words in planner order
                const double c0 = quad_bcast_d<0>(costV), c1v = quad_bcast_d<1>(costV), c2 = quad_bcast_d<2>(costV), c3 = quad_bcast_d<3>(costV);
                const double t0 = quad_bcast_d<0>(tw), t1 = quad_bcast_d<1>(tw), t2 = quad_bcast_d<2>(tw), t3 = quad_bcast_d<3>(tw);
                double bc = c0, bt = t0; bool right = false;
                { const bool lt = bc > c1v; bc = lt ? c1v : bc; bt = lt ? t1 : bt; right = lt ? true : right; }
                { const bool lt = bc > c2; bc = lt ? c2 : bc; bt = lt ? t2 : bt; right = lt ? false : right; }
                { const bool lt = bc > c3; bc = lt ? c3 : bc; bt = lt ? t3 : bt; right = lt ? true : right; }
                const bool ok = !flat && !(right ? (qi[4] - bt < pitchlims[0]) : (qi[4] + bt > pitchlims[1]));
                // (kmin differs between the lanes of a quad: any lane's objection sends the whole wavefront the literal way)
                if (!sca_dubins::lean::any_says(!flat && lean::keys_odd(kmin))) { len = bc; return ok; }
            } else if (!sca_dubins::lean::any_says(!flat && lean::keys_odd(kmin))) {                // (iii)
                QuadFrame V; V.F = F; V.mbeta = mbetaV;
                const Maneuver2D mv2 = plan2d_quad(V, qi[4], vr, sub, lane);
                bool okv = !flat && !(mv2.mode[0] == 'R' && mv2.mode[1] == 'L' && mv2.mode[2] == 'R');
                if (mv2.mode[0] == 'R') okv = okv && !(qi[4] - mv2.t < pitchlims[0]);
                else okv = okv && !(qi[4] + mv2.t > pitchlims[1]);
                len = mv2.length;
                return okv;
            }
        }
    }
    Maneuver2D mh, mv;
    const int nf = try_to_construct_quad(H, K, qi, qf, Rmin, pitchlims, hr, mh, mv, sub, lane);
    len = mv.length;
    return nf > 0;
}

// plan3d (sca_dubins.hpp) with the quad planner; the reference's three stages (first try, doubling, local search) are
// phases 0, 1, 2 of one loop so that the planner is inlined once
__device__ __forceinline__ sca_dubins::Plan3D plan3d_quad(const double qi[5], const double qf[5], double Rmin, const double pitchlims[2],
                                                          int sub, int lane) {
    using namespace sca_dubins;
    Plan3D P;
    double b = 1.0, step = 0.1, best = 0.0;
    const double qi2D[3] = {qi[0], qi[1], qi[3]}, qf2D[3] = {qf[0], qf[1], qf[3]};
    const QuadFrame H = frame2d_quad(qi2D, qf2D, sub);
    const SearchConst K = search_const(qi, qf, Rmin);
    const QuadWordSigns g = quad_word_signs(sub);
    const bool fast_ok = !sca_dubins::lean::any_says(!(Rmin >= 1e-40 && Rmin <= 1e40));
    int phase = 0, guard = 0;
    P.iters = 0;
    while (phase < 2 || ::fabs(step) > 1e-10) {
        double c;
        if (phase == 0) c = b;
        else if (phase == 1) { b *= 2.0; c = b; }
        else { c = b + step; if (c < 1.0) c = 1.0; }
        double lc;
        const bool fc = cand_quad(fast_ok, H, K, qi, qf, Rmin, pitchlims, Rmin * c, sub, lane, g, lc);
        P.iters++;
        if (phase < 2) {
            if (phase == 1 && ++guard > 200) return P;
            if (fc) { best = lc; phase = 2; }
            else phase = 1;
        } else {
            if (fc && lc < best) { b = c; best = lc; step *= 2.; }
            else step *= -0.1;
        }
    }
    Maneuver2D fbh, fbv;
    try_to_construct_quad(H, K, qi, qf, Rmin, pitchlims, Rmin * b, fbh, fbv, sub, lane);      // the winner's maneuvers
    const int it = P.iters;
    finish_plan(P, fbh, fbv, qi);
    P.iters = it;
    return P;
}

// ---- 16 / 32 / 64 lanes per plan: the search several candidates at a time --------------------------------------------------------
// The local search (dubinsmaneuver3d.py:86-100) is a chain: the next candidate radius is b + 2 step after a success, b - 0.1 step
// after a failure.  Both are known before the verdict on the current one, and so are their successors: the quads of a 16- (32-, 64-)
// lane group evaluate the current candidate and 2 (6, 14) of the candidates that can follow it, a TREE of verdict paths; the verdicts
// are then applied in the sequential loop's order along the path that loop would have taken -- same expressions for the candidates,
// same comparisons, results off the path are dropped -- until the path leaves the tree.
// WHICH tree only sets how far a round gets.  Until round 4 it was the balanced one (2 / 3 / 4 steps per round).  The chain is no coin
// toss, though: every search runs  F SSSSSS FF SSSSS FF SSSSSS FF ...  (an overshoot, the step back fails too, then five or six
// doublings to the next overshoot), so the trees of sca_spec_trees.h (generated: tools/gen_spec_trees.py, from the verdicts of the
// reference's own planner on 256 recorded searches) follow the likely continuations up to twelve steps deep, one tree per context =
// (kind of the current run of verdicts, its length, the lengths of the two runs before it): 6.6 - 7.5 steps per round with 15 candidates
// on held-out searches, 4.7 - 5.4 with 7, 2.6 - 2.7 with 3.  A quad finds its candidate by walking its node's path from the round's (b, step).
constexpr int TRK_SPEC2_MAX = 8192;        // <= this many re-plans in the pass: 3 candidates per round, 16 lanes per plan (2048 wavefronts: two per SIMD)
constexpr int TRK_SPEC3_MAX = 4096;        // <= this many: 7 per round, 32 lanes per plan (2048 wavefronts)
constexpr int TRK_SPEC4_MAX = 1024;        // <= this many: 15 per round, a whole wavefront per plan -- and a SIMD per wavefront (the kernel sits at the 256-register edge)

#if defined(SCA_KT_TIMING)   // debug builds: workgroup 0 / thread 0's clock over the pieces of a speculative search, summed over its rounds (tools/phase_clocks.py)
#define KG_ADD(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long t_ = wall_clock64(); sca_dubins::g_td_ticks[16 + (k)] += (int)(t_ - sca_dubins::g_td_last); sca_dubins::g_td_last = t_; } } while (0)
#define KG_ZERO() do { if (blockIdx.x == 0 && threadIdx.x == 0) { for (int k_ = 16; k_ < 32; k_++) sca_dubins::g_td_ticks[k_] = 0; sca_dubins::g_td_last = wall_clock64(); } } while (0)
#else
#define KG_ADD(k) do { } while (0)
#define KG_ZERO() do { } while (0)
#endif
template <int D> struct SpecTrees;
template <> struct SpecTrees<2> { static constexpr int TREES = sca_spec::TREES3, MAXD = sca_spec::MAXD3; static constexpr const uint8_t *of_ctx = sca_spec::TREE3_OF_CONTEXT; static constexpr const uint32_t *nodes = sca_spec::TREE3_NODES; };
template <> struct SpecTrees<3> { static constexpr int TREES = sca_spec::TREES7, MAXD = sca_spec::MAXD7; static constexpr const uint8_t *of_ctx = sca_spec::TREE7_OF_CONTEXT; static constexpr const uint32_t *nodes = sca_spec::TREE7_NODES; };
template <> struct SpecTrees<4> { static constexpr int TREES = sca_spec::TREES15, MAXD = sca_spec::MAXD15; static constexpr const uint8_t *of_ctx = sca_spec::TREE15_OF_CONTEXT; static constexpr const uint32_t *nodes = sca_spec::TREE15_NODES; };
// the trees of a kernel's form in LDS (a round reads its context's tree id and each quad its node; the walk reads through the lanes)
template <int D> struct SpecLds { uint2 nodes[SpecTrees<D>::TREES * (1 << D)]; uint8_t of_ctx[sca_spec::CONTEXTS]; };
template <int D> __device__ __forceinline__ void spec_trees_load(SpecLds<D> &S) {          // all threads, before the workgroup's first barrier
    for (int i = threadIdx.x; i < SpecTrees<D>::TREES * (1 << D); i += blockDim.x) S.nodes[i] = make_uint2(SpecTrees<D>::nodes[2 * i], SpecTrees<D>::nodes[2 * i + 1]);
    for (int i = threadIdx.x; i < sca_spec::CONTEXTS; i += blockDim.x) S.of_ctx[i] = SpecTrees<D>::of_ctx[i];
}

// a ballot's bits of the quads' first lanes (lanes 0, 4, 8, ...) as 16 contiguous bits
__device__ __forceinline__ unsigned quad_bits(unsigned long long m) {
    m &= 0x1111111111111111ull;
    m = (m | (m >> 3)) & 0x0303030303030303ull;
    m = (m | (m >> 6)) & 0x000f000f000f000full;
    m = (m | (m >> 12)) & 0x000000ff000000ffull;
    m = (m | (m >> 24)) & 0xffffull;
    return (unsigned)m;
}
// the (b, step) the sequential loop holds after the verdicts `bits` (bit i = 1: success), and the candidate it tries next
__device__ __forceinline__ void spec_path(unsigned bits, int plen, int maxd, double &nb, double &ns, double &c) {
#pragma unroll
    for (int i = 0; i < 12; i++) {
        if (i < maxd) {
            const bool on = i < plen, succ = on && ((bits >> i) & 1u);
            double cc = nb + ns;
            cc = cc < 1.0 ? 1.0 : cc;
            nb = succ ? cc : nb;                                         // success: b = c, step *= 2.; failure: step *= -0.1
            ns = ns * (on ? (succ ? 2. : -0.1) : 1.0);
        }
    }
    c = nb + ns;
    c = c < 1.0 ? 1.0 : c;
}

template <int D>
__device__ __forceinline__ sca_dubins::Plan3D plan3d_spec(const double qi[5], const double qf[5], double Rmin, const double pitchlims[2],
                                                          int sub, int lane, const SpecLds<D> &TR) {
    using namespace sca_dubins;
    constexpr int SLOTS = 1 << D, LANES = 4 << D, MAXD = SpecTrees<D>::MAXD;
    Plan3D P;
    const int quad = (lane & (LANES - 1)) >> 2, base = lane & ~(LANES - 1);
    const double qi2D[3] = {qi[0], qi[1], qi[3]}, qf2D[3] = {qf[0], qf[1], qf[3]};
    const QuadFrame H = frame2d_quad(qi2D, qf2D, sub);
    const SearchConst K = search_const(qi, qf, Rmin);
    const QuadWordSigns g = quad_word_signs(sub);
#if defined(SCA_QUAD_NO_FAST)                                           // measurement build: every candidate the literal way
    const bool fast_ok = false;
#else
    const bool fast_ok = !sca_dubins::lean::any_says(!(Rmin >= 1e-40 && Rmin <= 1e40));
#endif
    // The first round: the first try and the first doubling (:74-78) on quads 0 and 1 (b = 1, 2) -- and, since the doubling stage
    // ends at b = 2 for every pose of the benchmark configurations (33 of 40 in a +-20-m cube), the local search's first candidates
    // from (b = 2, step = 0.1) on the other quads: the nodes of the opening context's tree that fit.  A stage that does not end at
    // b = 2 goes on as the sequential loop does (all quads the same radius), and the speculated candidates are dropped.
    double b = 1.0, best_len = 0.0, step = 0.1;
    int ck = 0, cr = 0, cp = 0, cq = 0;                                  // the context: kind of the current run of verdicts (1 S, 2 F), its length, the two runs' before it
    KG_ADD(0);                                                           // frames and constants of the search
    // The walk along the sequential loop's path, all nodes at once (node k of the round's tree sits on quad QOFF + k).  A node ON
    // that path sees the best length the loop holds there: the length of the last node whose success its own path assumes (or the
    // round's incoming one) -- so every node can form the verdict it WOULD get; the path is then the set of nodes whose ancestors'
    // verdicts are the ones their paths assume (two ballots against the node's ancestor masks), and it ends at the node whose
    // verdict leads out of the tree (or to a candidate behind the loop's end).  That node's state and verdict are the round's
    // result: (b, step, best length), the count of candidates tried, the context.  (Until this form the walk went node by node
    // through v_readlane: 1.6 us of a 7-us round.)
    auto walk = [&](const int QOFF, const uint2 me, const double nb, const double ns, const double myc, const double mylen, const int nfc) {
        const int node = quad - QOFF, avail = SLOTS - 1 - (QOFF > 0 ? QOFF - 1 : 0);       // nodes 0 .. avail - 1 are on quads
        const unsigned pbits = me.x & 0xfffu;
        const int plen = (int)((me.x >> 12) & 15u);
        const bool valid = ::fabs(ns) > 1e-10;                           // the loop's condition in front of this candidate
        const int la = (int)((me.x >> 26) & 31u);
        // (fetched by every lane: a lane that skipped the fetch would be switched off while others read from it, and read as 0)
        const double seen_la = lane_fetch_d(mylen, base + 4 * (QOFF + (la > 0 ? la - 1 : 0)));
        const double seen = la > 0 ? seen_la : best_len;
        const bool acc = nfc > 0 && mylen < seen;
        const unsigned sh = (unsigned)(base >> 2) + (unsigned)QOFF, gm = (1u << avail) - 1u;
        const unsigned gA = (quad_bits(__ballot(acc)) >> sh) & gm, gV = (quad_bits(__ballot(valid)) >> sh) & gm;
        const unsigned am = me.y & 0xffffu, ab = me.y >> 16;
        const bool on = node >= 0 && node < avail && valid && (gA & am) == ab && (gV & am) == am;
        const int child = (int)((me.x >> (acc ? 16 : 21)) & 31u);
        const bool last = on && !(child > 0 && child <= avail && ((gV >> (child - 1)) & 1u));
        const unsigned gL = (quad_bits(__ballot(last)) >> sh) & gm;       // exactly one node (were it none, the root alone: its verdict always holds)
        const int src = base + 4 * (QOFF + (gL ? __ffs((int)gL) - 1 : 0));
        double Lb, Ls, Lc, Llen, Lseen;
        int Lacc, Lplen;
        unsigned Lbits;
        if constexpr (LANES == 64) {
            const int sl = __builtin_amdgcn_readfirstlane(src);
            Lb = readlane_f64(nb, sl); Ls = readlane_f64(ns, sl); Lc = readlane_f64(myc, sl); Llen = readlane_f64(mylen, sl); Lseen = readlane_f64(seen, sl);
            Lacc = __builtin_amdgcn_readlane((int)acc, sl); Lplen = __builtin_amdgcn_readlane(plen, sl); Lbits = (unsigned)__builtin_amdgcn_readlane((int)pbits, sl);
        } else {
            Lb = lane_fetch_d(nb, src); Ls = lane_fetch_d(ns, src); Lc = lane_fetch_d(myc, src); Llen = lane_fetch_d(mylen, src); Lseen = lane_fetch_d(seen, src);
            Lacc = __shfl((int)acc, src); Lplen = __shfl(plen, src); Lbits = (unsigned)__shfl((int)pbits, src);
        }
        if (Lacc) { b = Lc; best_len = Llen; step = Ls * 2.; }
        else { b = Lb; best_len = Lseen; step = Ls * -0.1; }
        P.iters += Lplen + 1;
        const unsigned verdicts = Lbits | ((unsigned)Lacc << Lplen);       // of the Lplen + 1 candidates the loop has tried in this round
#pragma unroll 1
        for (int i = 0; i <= Lplen; i++) {
            const int kind = ((verdicts >> i) & 1u) ? 1 : 2;
            if (kind == ck) cr++;
            else { cq = cp; cp = cr; cr = 1; ck = kind; }
        }
    };
    // The first round: the first try and the first doubling (:74-78) on quads 0 and 1 (b = 1, 2) -- and, since the doubling stage
    // ends at b = 2 for every pose of the benchmark configurations (33 of 40 in a +-20-m cube), the local search's first candidates
    // from (b = 2, step = 0.1) on the other quads: the nodes of the opening context's tree that fit.  A stage that does not end at
    // b = 2 goes on as the sequential loop does (all quads the same radius), and the speculated candidates are dropped.
    {
        constexpr int DQ = 2;
        const int tree0 = TR.of_ctx[0];
        const uint2 me = quad >= DQ ? TR.nodes[tree0 * SLOTS + (quad - DQ)] : make_uint2(0u, 0u);
        double nb = 2.0, ns = 0.1, myc;
        spec_path(me.x & 0xfffu, (int)((me.x >> 12) & 15u), MAXD, nb, ns, myc);
        myc = quad == 0 ? 1.0 : (quad == 1 ? 2.0 : myc);
        double mylen;
        KG_ADD(1);
        const int nfc = cand_quad(fast_ok, H, K, qi, qf, Rmin, pitchlims, Rmin * myc, sub, lane, g, mylen) ? 2 : 0;
        KG_ADD(2);
        const int nf0 = __shfl(nfc, base), nf1 = __shfl(nfc, base + 4);
        const double len0 = lane_fetch_d(mylen, base), len1 = lane_fetch_d(mylen, base + 4);
        P.iters = 1;
        if (nf0 > 0) best_len = len0;                                    // b = 1 is feasible: no doubling; the search starts from there
        else {
            b = 2.0;
            P.iters = 2;
            if (nf1 > 0) {
                best_len = len1;
                walk(DQ, me, nb, ns, myc, mylen, nfc);
            } else {
                bool fb = false;
                int guard = 1;                                           // (b = 2 was the first doubling)
                while (!fb) {
                    b *= 2.0;
                    fb = cand_quad(fast_ok, H, K, qi, qf, Rmin, pitchlims, Rmin * b, sub, lane, g, best_len);
                    P.iters++;
                    if (++guard > 200) return P;
                }
            }
        }
    }
    KG_ADD(3);
    while (::fabs(step) > 1e-10) {
        KG_ADD(7);
#if defined(SCA_SPEC_BALANCED)                                          // measurement build: the balanced tree in every round
        const int tree = 0;
#else
        const int tree = TR.of_ctx[((ck * (sca_spec::RUN_CAP + 1) + (cr < sca_spec::RUN_CAP ? cr : sca_spec::RUN_CAP)) * (sca_spec::PREV_CAP + 1) +
                                    (cp < sca_spec::PREV_CAP ? cp : sca_spec::PREV_CAP)) * (sca_spec::PREV2_CAP + 1) + (cq < sca_spec::PREV2_CAP ? cq : sca_spec::PREV2_CAP)];
#endif
        const uint2 me = TR.nodes[tree * SLOTS + quad];                  // (the spare quad's words are 0: the root again, counted out by the walk)
        // this quad's candidate: the (b, step) the sequential loop holds when it has taken the node's path
        double nb = b, ns = step, myc;
        spec_path(me.x & 0xfffu, (int)((me.x >> 12) & 15u), MAXD, nb, ns, myc);
        double mylen;
        KG_ADD(4);
        const int nfc = cand_quad(fast_ok, H, K, qi, qf, Rmin, pitchlims, Rmin * myc, sub, lane, g, mylen) ? 2 : 0;
        KG_ADD(5);
#if defined(SCA_KT_TIMING)
        if (blockIdx.x == 0 && threadIdx.x == 0) sca_dubins::g_td_ticks[31]++;
#endif
        walk(0, me, nb, ns, myc, mylen, nfc);
        P.rounds++;
        KG_ADD(6);
    }
    Maneuver2D fbh, fbv;
    try_to_construct_quad(H, K, qi, qf, Rmin, pitchlims, Rmin * b, fbh, fbv, sub, lane);      // the winner's maneuvers
    KG_ADD(8);
    const int it = P.iters, rd = P.rounds + 1;                           // (+ the opening round)
    finish_plan(P, fbh, fbv, qi);
    P.iters = it; P.rounds = rd;
    KG_ADD(9);
    return P;
}

// The per-agent form of agent.turning_radius / agent.pitchlims (sca_device_tracker_set_agent_params with more classes than launches are
// worth): a wavefront that plans ONE agent loads that agent's three values through a wave-uniform index, i.e. into scalar registers like
// the kernel arguments they replace -- the search keeps its register allocation.  T.plo_pa is null in every other form.
__device__ __forceinline__ void own_plan_params(const sca_dubins::TrackView &T, int agent, double &Rmin, double pl[2]) {
    if (!T.plo_pa) return;
    const int a = __builtin_amdgcn_readfirstlane(agent);
    Rmin = T.R_pa[a]; pl[0] = T.plo_pa[a]; pl[1] = T.phi_pa[a];
}

// one group of LANES lanes per plan: 4 = the quad planner, 16 / 32 / 64 = speculative search of depth 2 / 3 / 4
template <int LANES, typename TREES>
__device__ __forceinline__ void replan_group(const DeviceView d, const sca_dubins::TrackView T, const TrackDev K, int count, const TREES &TR) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int idx = gid / LANES, sub = gid & 3, lane = threadIdx.x & 63;
    if (idx >= count) return;                                            // whole groups leave together
    KG_ZERO();
    const int agent = trk_list_agent(K.list, K.bcount + K.parity * TRK_BUCKETS, K.n, idx);
    if (T.cls && T.cls[agent] != T.class_id) return;                       // (whole groups: another class's launch plans this agent)
    const PubRec r = d.rec[agent];
    const double pos[3] = {r.px, r.py, r.pz};
    const double heading[3] = {d.heading[agent * 3], d.heading[agent * 3 + 1], d.heading[agent * 3 + 2]};
    double qi[5], qf[5];
    sca_dubins::dubins_endpoints(T, agent, pos, heading, qi, qf);
    double pl[2] = {T.pitch_lo, T.pitch_hi};
    sca_dubins::Plan3D P;
    if constexpr (LANES == 4) P = plan3d_quad(qi, qf, T.turning_radius, pl, sub, lane);
    else if constexpr (LANES == 16) P = plan3d_spec<2>(qi, qf, T.turning_radius, pl, sub, lane, TR);
    else if constexpr (LANES == 32) P = plan3d_spec<3>(qi, qf, T.turning_radius, pl, sub, lane, TR);
    else {
        double Rmin = T.turning_radius;
        own_plan_params(T, agent, Rmin, pl);                              // (the per-agent form: this wavefront's agent's own three values)
        P = plan3d_spec<4>(qi, qf, Rmin, pl, sub, lane, TR);
    }
    if ((gid & (LANES - 1)) != 0) return;
    sca_dubins::AgentTrack &a = K.st[agent];
    double dif[3], V[3];
    sca_dubins::track_adopt(a, P, pos, dif);
    sca_dubins::track_finish(T, a, agent, pos, dif, V);
    track_store(d, K, agent, V);
    KG_ADD(10);
}

// The many-lanes-per-plan forms, ONE KERNEL EACH (round 2; they used to share one kernel that picked the form by the count:
// its register allocation was the union's -- 258, one wavefront per SIMD -- and moved with every change to any form; alone the
// quad form takes 217 registers and runs two wavefronts per SIMD, which is what 16 385 .. 32 768 plans need).  Every launch
// reads the pass's count and returns unless it falls into its range (lo, hi] (launch_tracker).
constexpr int TRK_GROUP_THREADS = 256;
template <int LANES>
__global__ __launch_bounds__(TRK_GROUP_THREADS, LANES == 4 ? 2 : 1) void k_replan_group(DeviceView d, sca_dubins::TrackView T, TrackDev K) {
    SCA_TL(d, TL_REPLAN);
    if constexpr (LANES == 4) {
        sca_gm::lds_tables_load();                                       // atan2's and sin / cos's tables into LDS (all threads, first)
        const int count = K.count[K.parity];
        if (count <= K.lo || count > K.hi) return;
        replan_group<LANES>(d, T, K, count, 0);
    } else {
        constexpr int D = LANES == 16 ? 2 : (LANES == 32 ? 3 : 4);
        __shared__ SpecLds<D> TR;
        spec_trees_load<D>(TR);                                          // (the barrier that ends lds_tables_load covers them)
        sca_gm::lds_tables_load();
        const int count = K.count[K.parity];
        if (count <= K.lo || count > K.hi) return;
        replan_group<LANES>(d, T, K, count, TR);
    }
}

// k_track + k_replan_group<64> in ONE launch for shards of so few agents that every AGENT can have a wavefront (round 4): the
// wavefront's first lane takes the follow-or-re-plan decision (k_track's chain of restated-libm calls, ~10 us, which a launch of its own
// only pads with its tables and its gap), then the whole wavefront searches if the agent re-plans -- no list, no order by expected length
// (every plan has a SIMD of its own anyway).  Counts the re-plans for the host's choice of forms as k_track does.
__global__ __launch_bounds__(TRK_GROUP_THREADS, 1) void k_track_group(DeviceView d, sca_dubins::TrackView T, TrackDev K) {
    SCA_TL(d, TL_REPLAN);
    __shared__ SpecLds<4> TR;
    spec_trees_load<4>(TR);                                              // (the barrier that ends lds_tables_load covers them)
    sca_gm::lds_tables_load();
    if (blockIdx.x == 0 && threadIdx.x == 0) K.count[(K.parity + 1) & 3] = 0;
    if (blockIdx.x == 0 && threadIdx.x < TRK_BUCKETS) K.bcount[((K.parity + 1) & 3) * TRK_BUCKETS + threadIdx.x] = 0;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int idx = gid >> 6, sub = gid & 3, lane = threadIdx.x & 63;
    if (idx >= shard_size(d)) return;                                    // whole wavefronts leave together
    const int agent = shard_agent(d, idx);
    if (!track_active(d, agent)) return;
    if (T.cls && T.cls[agent] != T.class_id) return;                     // (decision AND search by the agent's own class's launch)
    const PubRec r = d.rec[agent];
    const double pos[3] = {r.px, r.py, r.pz};
    sca_dubins::AgentTrack &a = K.st[agent];
    double dif[3] = {0, 0, 0}, V[3];
    int replan = 0;
    if (lane == 0) {
        const float vel[3] = {r.vx, r.vy, r.vz};
        replan = sca_dubins::track_decide(T, a, agent, pos, vel, K.nbr0[agent], dif) ? 1 : 0;
        if (!replan) {
            sca_dubins::track_finish(T, a, agent, pos, dif, V);
            track_store(d, K, agent, V);
        } else atomicAdd(&K.count[K.parity], 1);
    }
    if (__builtin_amdgcn_readfirstlane(replan) == 0) return;
    const double heading[3] = {d.heading[agent * 3], d.heading[agent * 3 + 1], d.heading[agent * 3 + 2]};
    double qi[5], qf[5];
    sca_dubins::dubins_endpoints(T, agent, pos, heading, qi, qf);
    double pl[2] = {T.pitch_lo, T.pitch_hi};
    double Rmin = T.turning_radius;
    own_plan_params(T, agent, Rmin, pl);                                 // (the per-agent form: this wavefront's agent's own three values)
    const sca_dubins::Plan3D P = plan3d_spec<4>(qi, qf, Rmin, pl, sub, lane, TR);
    if (lane != 0) return;
    sca_dubins::track_adopt(a, P, pos, dif);
    sca_dubins::track_finish(T, a, agent, pos, dif, V);
    track_store(d, K, agent, V);
}

// self-test: the device build of sca_glibc_math.h, one function per launch (fn as sca_selftest_libm numbers them: 0-4 the
// branch-free forms the kernels use, 5-8 the literal restatements, 9 / 10 the two results of the fused sincos, 11-14 the
// constant-table entry points of the policy epilogue and the env update: atan2, sin, cos, pow(x, 2))
__global__ __launch_bounds__(256) void k_selftest_libm(int fn, const double *a, const double *b, int n, double *out) {
    sca_gm::lds_tables_load();                                           // atan2's and sin / cos's tables into LDS (all threads, first)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r;
    double s2, c2;
    switch (fn) {
    case 0: r = sca_gm::g_sin(a[i]); break;
    case 1: r = sca_gm::g_cos(a[i]); break;
    case 2: r = sca_gm::g_acos(a[i]); break;
    case 3: r = sca_gm::g_atan2(a[i], b[i]); break;
    case 4: r = sca_gm::g_pow2(a[i]); break;
    case 5: r = sca_gm::g_sin_ref(a[i]); break;
    case 6: r = sca_gm::g_cos_ref(a[i]); break;
    case 7: r = sca_gm::g_atan2_ref(a[i], b[i]); break;
    case 8: r = sca_gm::g_pow2_ref(a[i]); break;
    case 9: sca_gm::g_sincos(a[i], s2, c2); r = s2; break;
    case 10: sca_gm::g_sincos(a[i], s2, c2); r = c2; break;
    case 11: r = m_atan2(a[i], b[i]); break;                             // what cartesian2spherical / get_phi / update_velocitie call:
    case 12: m_sincos(a[i], s2, c2); r = s2; break;                      // the same forms on the constant tables (sca_core.h)
    case 13: m_sincos(a[i], s2, c2); r = c2; break;
    default: r = m_pow2(a[i]); break;
    }
    out[i] = r;
}

__global__ __launch_bounds__(256) void k_track_replans(const sca_dubins::AgentTrack *st, int32_t *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = st[i].replans;
}

}  // namespace sca
