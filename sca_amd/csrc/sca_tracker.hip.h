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
//             dense wavefronts keeps the (rare, long) plans from stalling the 63 followers that would share their wave.
//
// The arithmetic is sca_dubins.hpp compiled for gfx950: same statements as the host tracker, the device library's
// sin / cos / atan2 / acos instead of glibc's.  State: one AgentTrack record per agent, resident in HBM.
#pragma once
#include <hip/hip_runtime.h>
#include "sca_dubins.hpp"
#include "sca_kernels.hip.h"

namespace sca {

struct TrackDev {
    sca_dubins::AgentTrack *st;   // [n]
    double *nbr0;                 // [n] distSq of agent.neighbors[0] as the previous pass left it (-1: empty list)
    int32_t *list;                // [n] agents that re-plan in this pass
    int32_t *count;               // [2] list length, double-buffered by pass parity (k_track zeroes the other one)
    int parity;
    int nbr0_from_lists;          // 1: refresh nbr0 from the neighbour lists of the previous pass (resident stepping)
};

constexpr int TRK_REPLAN_LANES = 64;      // one wavefront per workgroup: re-plans spread over as many CUs as possible

__device__ __forceinline__ bool track_active(const DeviceView &d, int agent) {
    const int pol = d.policy[agent];
    return (pol == POL_SCA || pol == POL_RVO_DUBINS) && (d.rec[agent].flags & 7u) == 0u;   // mampenv.py:35
}
__device__ __forceinline__ void track_store(const DeviceView &d, int agent, const double *V) {
    for (int q = 0; q < 3; q++) {
        double x = V[q];
        if (x != x) x = 0.0;                                             // what numpy.nan_to_num does on the host path
        else if (x > 1.7976931348623157e308) x = 1.7976931348623157e308;
        else if (x < -1.7976931348623157e308) x = -1.7976931348623157e308;
        d.vpref_ext[agent * 3 + q] = x;
    }
}

__global__ __launch_bounds__(256) void k_track(DeviceView d, sca_dubins::TrackView T, TrackDev K) {
    if (blockIdx.x == 0 && threadIdx.x == 0) K.count[K.parity ^ 1] = 0;
    const int agent = d.shard_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (agent >= d.shard_begin + d.shard_count) return;
    double nb0 = K.nbr0[agent];
    if (K.nbr0_from_lists && d.nbr_valid[agent]) {                       // lists of the previous pass (agent.py:79-99)
        nb0 = d.nbr_n[agent] > 0 ? d.nbr_dsq[(size_t)agent * K_MAX] : -1.0;
        K.nbr0[agent] = nb0;
    }
    if (!track_active(d, agent)) return;
    const PubRec r = d.rec[agent];
    const double pos[3] = {r.px, r.py, r.pz};
    const float vel[3] = {r.vx, r.vy, r.vz};
    sca_dubins::AgentTrack &a = K.st[agent];
    double dif[3], V[3];
    const bool replan = sca_dubins::track_decide(T, a, agent, pos, vel, nb0, dif);
    if (replan) {
        // wave-aggregated append
        const unsigned long long m = __ballot(1);
        const int lane = threadIdx.x & 63;
        const int leader = __ffsll((long long)m) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&K.count[K.parity], __popcll(m));
        base = __shfl(base, leader);
        K.list[base + __popcll(m & ((1ull << lane) - 1ull))] = agent;
        return;
    }
    sca_dubins::track_finish(T, a, agent, pos, dif, V);
    track_store(d, agent, V);
}

__global__ __launch_bounds__(TRK_REPLAN_LANES, 2) void k_replan(DeviceView d, sca_dubins::TrackView T, TrackDev K) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K.count[K.parity]) return;
    const int agent = K.list[idx];
    const PubRec r = d.rec[agent];
    const double pos[3] = {r.px, r.py, r.pz};
    const double heading[3] = {d.heading[agent * 3], d.heading[agent * 3 + 1], d.heading[agent * 3 + 2]};
    sca_dubins::AgentTrack &a = K.st[agent];
    double dif[3], V[3];
    sca_dubins::track_replan(T, a, agent, pos, heading, dif);
    sca_dubins::track_finish(T, a, agent, pos, dif, V);
    track_store(d, agent, V);
}

__global__ __launch_bounds__(256) void k_track_replans(const sca_dubins::AgentTrack *st, int32_t *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = st[i].replans;
}

}  // namespace sca
