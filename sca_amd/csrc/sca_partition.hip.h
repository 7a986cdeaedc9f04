// sca_partition.hip.h -- cell-owner partition of SCA_NBR_GRID with halo exchange (the second half).
//
// With the all-gather every rank receives and bins all N public records every step.  Here space is cut into
// slabs of grid cells along one axis; a rank OWNS the agents whose cell lies in its slab and additionally holds copies of the
// agents in the one layer of cells on either side of it (the halo): the 27 cells around any of its agents -- all the neighbour
// query (k_neighbors_grid: kdTree.py:124-156 / agent.py:79-99 on the grid) and the collision check (mampenv.py:61-80) ever
// look at -- are then present.  Per step a rank sends its two slab neighbours
//   * the old and the moved public record of every agent in its layer of cells next to that neighbour (96 bytes + id), and
//   * agents that crossed the cut, with their private state (heading, v_pref, distances, the tracker record): ownership
//     moves with the agent,
// instead of receiving N records: the replicated O(N) part of a step is gone (bench figures in DESIGN.md section 6).
//
// The arrays stay indexed by global agent id on every rank (the static per-agent inputs of sca_set_agents are replicated as
// before; only rows of present agents are ever read), so the kernels of the hot path are the ones of the contiguous-shard
// mode: they take their i-th agent from DeviceView::own instead of shard_begin + i (shard_agent()), the grid is built over
// DeviceView::present = owned + halo.  Results are those of a single rank bit for bit: the grid's lists are ordered by
// (distSq, obstacle first, id), i.e. independent of which rank holds what, and every pairwise collision test is evaluated by
// the owner of either side from the same two pairs of records (tests/test_gpu_partition.py).
//
// One step of a rank:  sca_step_begin (grid over present, policy pass + integrate for owned)
//                      -> sca_partition_pack(side) x 2 -> [exchange with the slab neighbours] -> sca_partition_unpack(side) x 2
//                      -> sca_partition_commit (ownership moves; immigrants join before the collision check: their new owner
//                         checks them by traversal, the near lists of the pass stayed with the old owner)
//                      -> sca_step_end (collision / goal flags for owned, at-goal for the halo copies).
#pragma once
#include <hip/hip_runtime.h>
#include "sca_kernels.hip.h"
#include "sca_grid.hip.h"
#include "sca_dubins.hpp"

namespace sca {

struct PartHalo { int32_t id, pad; PubRec old_rec, new_rec; };                  // 104 bytes
struct PartMig {                                                               // an agent that changes owner
    int32_t id, step_num;
    double heading[3], vpref_ext[3], total_dist, nbr0;
    // what the step left for the caller (sca_get_actions / sca_get_diag): the agent's row travels with it
    float action[8];
    double vpref_used[3];
    int32_t diag[8], status, pad;
    sca_dubins::AgentTrack trk;
};
struct PartHeader { int32_t n_halo, n_mig, cap_halo, cap_mig; };

struct PartDev {
    int32_t *present[2];      // [n] owned agents first, then the halo copies; [cur] is what the kernels see, [cur ^ 1] is being built
    int32_t *halo_tmp;        // [n] the halo copies of the lists being built, until the owned part is complete
    int32_t *counts;          // [16]: 0 owned, 1 halo (current lists); 2 owned, 3 halo (lists being built); 4 overflow flag;
                              // 5 .. 8 halo / migrant counters of the two outgoing messages; 9 workgroups done (last-one-out)
    uint8_t *emig;            // [n] 0, or 1 + side the agent leaves to in this step
    int cur;
    int axis;                 // 0 x, 1 y, 2 z
    long long lo_cell, hi_cell;   // this rank's slab: cells [lo_cell, hi_cell) along the axis (LLONG_MIN / LLONG_MAX at the ends)
    int has_peer[2];          // a slab neighbour below / above
    int cap_halo, cap_mig;    // entries per message
    double inv_cell;
    sca_dubins::AgentTrack *trk_st;   // the device tracker's records (null: no tracker)
    double *trk_nbr0;
};

__device__ __forceinline__ long long part_cell(const PartDev &p, const PubRec &r) {
    return grid_cell(p.axis == 0 ? r.px : (p.axis == 1 ? r.py : r.pz), p.inv_cell);
}
__host__ __device__ inline size_t part_message_bytes(int cap_halo, int cap_mig) {
    return sizeof(PartHeader) + (size_t)cap_halo * sizeof(PartHalo) + (size_t)cap_mig * sizeof(PartMig);
}

// append `value` to list[*counter ...] for the lanes with `want`: one atomic per wavefront (every owned agent appends every step,
// and same-address atomics serialise at the L2, ~12 ns each: 50 000 of them were 0.6 ms)
__device__ __forceinline__ void wave_append(bool want, int32_t *list, int32_t *counter, int value) {
    const unsigned long long m = __ballot(want);
    if (!m) return;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader);
    if (want) list[base + __popcll(m & ((1ull << lane) - 1ull))] = value;
}

// the initial lists, from a state every rank holds completely (sca_set_state): owned = in my slab, halo = in the adjacent layers
__global__ __launch_bounds__(256) void k_part_init(DeviceView d, PartDev p) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    bool mine = false;
    if (a < d.n) {
        p.emig[a] = 0;
        const long long c = part_cell(p, d.rec[a]);
        mine = c >= p.lo_cell && c < p.hi_cell;
    }
    wave_append(mine, p.present[p.cur ^ 1], &p.counts[2], a);
}
__global__ __launch_bounds__(256) void k_part_init_halo(DeviceView d, PartDev p) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    bool h = false;
    if (a < d.n) {
        const long long c = part_cell(p, d.rec[a]);
        h = (p.has_peer[0] && c == p.lo_cell - 1) || (p.has_peer[1] && c == p.hi_cell);
    }
    wave_append(h, p.halo_tmp, &p.counts[3], a);
}

// what goes to the two slab neighbours (buf[0]: below, buf[1]: above; null: no neighbour there): one lane per owned agent.  The
// message counters live in PartDev::counts[5 .. 8] while the kernel runs; the workgroup that finishes last writes them into the
// message headers and clears them for the next step (no memset launches around the kernel).
__global__ __launch_bounds__(256) void k_part_pack(DeviceView d, PartDev p, uint8_t *buf0, uint8_t *buf1) {
    __shared__ int last;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < shard_size(d)) {
        const int a = d.own[i];
        const PubRec ro = d.rec[a], rn = d.rec_new[a];
        const long long co = part_cell(p, ro), cn = part_cell(p, rn);
#pragma unroll
        for (int side = 0; side < 2; side++) {
            uint8_t *buf = side ? buf1 : buf0;
            if (!buf) continue;
            const long long edge = side == 0 ? p.lo_cell : p.hi_cell - 1;    // my layer of cells next to that neighbour
            const bool leaves = side == 0 ? cn < p.lo_cell : cn >= p.hi_cell;
            if (!(co == edge || cn == edge || leaves)) continue;
            PartHalo *halo = (PartHalo *)(buf + sizeof(PartHeader));
            PartMig *mig = (PartMig *)(buf + sizeof(PartHeader) + (size_t)p.cap_halo * sizeof(PartHalo));
            const int s = atomicAdd(&p.counts[5 + 2 * side], 1);
            if (s < p.cap_halo) { PartHalo e; e.id = a; e.pad = leaves ? 1 : 0; e.old_rec = ro; e.new_rec = rn; halo[s] = e; }
            else atomicOr(&p.counts[4], 1);
            if (leaves) {
                p.emig[a] = (uint8_t)(1 + side);
                const int m = atomicAdd(&p.counts[6 + 2 * side], 1);
                if (m < p.cap_mig) {
                    PartMig &e = mig[m];
                    e.id = a; e.step_num = d.step_num[a];
                    for (int q = 0; q < 3; q++) { e.heading[q] = d.heading[a * 3 + q]; e.vpref_ext[q] = d.vpref_ext[a * 3 + q]; }
                    e.total_dist = d.total_dist[a];
                    for (int q = 0; q < 8; q++) { e.action[q] = d.action[(size_t)a * 8 + q]; e.diag[q] = d.diag[(size_t)a * 8 + q]; }
                    for (int q = 0; q < 3; q++) e.vpref_used[q] = d.vpref_used[a * 3 + q];
                    e.status = d.status[a]; e.pad = 0;
                    e.nbr0 = p.trk_nbr0 ? p.trk_nbr0[a] : -1.0;
                    if (p.trk_st) e.trk = p.trk_st[a];
                } else atomicOr(&p.counts[4], 1);
            }
        }
    }
    // the last workgroup publishes the counters.  No fence: it reads nothing but the counters, which only atomics touch -- every
    // lane holds the result of its own before it reaches the barrier -- and the entries are visible to the next kernel anyway.
    // (A device-scope fence here is a write-back of the XCD's L2: 15 us per launch, measured.)
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&p.counts[9], 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (last && threadIdx.x == 0) {
        for (int side = 0; side < 2; side++) {
            uint8_t *buf = side ? buf1 : buf0;
            const int nh = atomicExch(&p.counts[5 + 2 * side], 0), nm = atomicExch(&p.counts[6 + 2 * side], 0);
            if (buf) { PartHeader *H = (PartHeader *)buf; H->n_halo = nh; H->n_mig = nm; H->cap_halo = p.cap_halo; H->cap_mig = p.cap_mig; }
        }
        p.counts[9] = 0;
    }
}

// the neighbours' messages (null: none): their records become my halo copies (or my agents, when they crossed over)
__global__ __launch_bounds__(256) void k_part_unpack(DeviceView d, PartDev p, const uint8_t *buf0, const uint8_t *buf1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int side = 0; side < 2; side++) {
        const uint8_t *buf = side ? buf1 : buf0;
        if (!buf) continue;
        const PartHeader *H = (const PartHeader *)buf;
        const PartHalo *halo = (const PartHalo *)(buf + sizeof(PartHeader));
        const PartMig *mig = (const PartMig *)(buf + sizeof(PartHeader) + (size_t)p.cap_halo * sizeof(PartHalo));
        const int nh = H->n_halo < p.cap_halo ? H->n_halo : p.cap_halo, nm = H->n_mig < p.cap_mig ? H->n_mig : p.cap_mig;
        if (H->n_halo > p.cap_halo || H->n_mig > p.cap_mig) { if (i == 0) atomicOr(&p.counts[4], 2); }
        bool keep = false;
        int hid = 0;
        if (i < nh) {
            const PartHalo e = halo[i];
            d.rec[e.id] = e.old_rec;
            d.rec_new[e.id] = e.new_rec;
            const long long cn = part_cell(p, e.new_rec);
            keep = cn == p.lo_cell - 1 || cn == p.hi_cell;
            hid = e.id;
        }
        wave_append(keep, p.halo_tmp, &p.counts[3], hid);
        const bool im = i < nm;
        int a = 0;
        if (im) {
            const PartMig &e = mig[i];
            a = e.id;
            d.step_num[a] = e.step_num;
            for (int q = 0; q < 3; q++) { d.heading[a * 3 + q] = e.heading[q]; d.vpref_ext[a * 3 + q] = e.vpref_ext[q]; }
            d.total_dist[a] = e.total_dist;
            for (int q = 0; q < 8; q++) { d.action[(size_t)a * 8 + q] = e.action[q]; d.diag[(size_t)a * 8 + q] = e.diag[q]; }
            for (int q = 0; q < 3; q++) d.vpref_used[a * 3 + q] = e.vpref_used[q];
            d.status[a] = e.status;
            if (p.trk_nbr0) p.trk_nbr0[a] = e.nbr0;
            if (p.trk_st) p.trk_st[a] = e.trk;
            d.near_n[a] = -1;                                 // the pass's collision candidates stayed with the old owner: traverse
        }
        wave_append(im, p.present[p.cur ^ 1], &p.counts[2], a);
    }
}

// my own agents into the lists being built: those that stay are owned, those that left are halo copies now
__global__ __launch_bounds__(256) void k_part_keep(DeviceView d, PartDev p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && (shard_size(d) > d.shard_count || present_count(d) > d.n_present)) atomicOr(&p.counts[4], 8);   // the host's bounds were too small
    const bool live = i < shard_size(d);
    const int a = live ? d.own[i] : 0;
    const bool left = live && p.emig[a] != 0;
    if (left) p.emig[a] = 0;
    wave_append(left, p.halo_tmp, &p.counts[3], a);
    wave_append(live && !left, p.present[p.cur ^ 1], &p.counts[2], a);
}
// ... and, in a launch of its own (the copies other workgroups appended must be visible: a kernel boundary is the cheap way):
// the halo copies behind the owned agents; SHUFFLE: the new counts in place of the current ones, accumulators cleared
template <bool SHUFFLE>
__global__ __launch_bounds__(256) void k_part_close(DeviceView d, PartDev p) {
    const int no = p.counts[2], nh = p.counts[3];
    if (no + nh > d.n) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&p.counts[4], 4); return; }
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nh; k += gridDim.x * blockDim.x) p.present[p.cur ^ 1][no + k] = p.halo_tmp[k];
    if (SHUFFLE) {
        // every workgroup has read the accumulators above before the last one to arrive moves them
        __shared__ int last;
        __syncthreads();
        if (threadIdx.x == 0) last = atomicAdd(&p.counts[9], 1) == (int)gridDim.x - 1;
        __syncthreads();
        if (last && threadIdx.x == 0) { p.counts[0] = no; p.counts[1] = nh; p.counts[2] = 0; p.counts[3] = 0; p.counts[9] = 0; }
    }
}

}  // namespace sca
