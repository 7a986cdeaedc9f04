// sca_kdbuild.hip.h -- K0: the reference's per-step agent kd-tree, rebuilt ON THE DEVICE.
//
// Replica of KDTree.buildAgentTreeRecursive (kdTree.py:60-122): same nodes, same boxes, same split planes and --
// because the neighbour list of agent.py:79-99 depends on the visit order -- the SAME permutation of agentIDs,
// including its history dependence (the permutation is never reset, kdTree.py:43-45).
//
// The sequential Hoare-style partition of kdTree.py:101-111 (two pointers walking inwards, swapping the first
// element >= split found from the left with the last element < split found from the right) is equivalent to:
//     L = #(elements < split);  the k-th element >= split inside [b, b+L)  (counted from the left)
//     swaps with                the k-th element <  split inside [b+L, e)  (counted from the right);
//     every other element stays where it is.
// That form is data-parallel: one prefix count per element.
//
//   k_kd_gather : coordinates into position order (kx/ky/kz[p] = pos of ids[p]) -> every later pass is coalesced;
//                 root box; the per-agent prologue of the solver for the rank's shard
//   k_kd_lv_*   : two launches per tree level for the nodes with more than wave_max members; a node is cut into
//                 chunks of KD_CHUNK positions, one workgroup per chunk: (flags + chained scan + ranks + children's
//                 boxes) -> (swaps + query record + children and their chunk records)
//   k_kd_top    : (trees of <= 4096 members, round 4) the nodes with more than wave_max members by ONE workgroup in LDS instead of
//                 the level passes
//   k_kd_block  : every subtree of <= wave_max members is finished by ONE WORKGROUP entirely in LDS, level by level,
//                 all nodes of a level at once (element-parallel; boxes as DPP minima of order-preserving keys, LDS
//                 atomics for what is left, one lane per node for the records)
#pragma once
#include "sca_kernels.hip.h"

namespace sca {
// The kd build is a chain of ~17 short dependent launches; in a tracked pass it runs beside the re-plan kernel, whose wavefronts are
// older and dense in VALU work, and the SIMD's arbiter serves the oldest wavefront first: the build's wavefronts got the issue
// slots that were left (k_kd_block 124 us against 33 alone, a level pass 18.5 against 9.1).  With s_setprio they go first on the
// SIMDs they share -- a few thousand instructions each -- and the re-plan kernel does not get measurably longer: k_kd_block 56 us,
// level pass 11.6, c4 step 0.759 -> 0.743 ms.  (SCA_KD_PRIO=0 at build time switches it off.)
#ifndef SCA_KD_PRIO
#define SCA_KD_PRIO 3
#endif
#if SCA_KD_PRIO > 0
#define SCA_KD_SETPRIO() __builtin_amdgcn_s_setprio(SCA_KD_PRIO)
#else
#define SCA_KD_SETPRIO() ((void)0)
#endif

// A node of <= wave_max members (KdScratch::wave_max, chosen per build) is finished, whole subtree, by ONE WORKGROUP in LDS
// (k_kd_block).  The host picks it between these bounds so that the node sizes of a level (n / 2^k, within a few per cent)
// do not straddle it: with a fixed 1024 the 4096- and 16384-agent trees needed a whole level pass for the half of their
// ~1024-member nodes that were a little larger.
constexpr int KD_WAVE_MIN = 768, KD_WAVE_CAP = 1536;   // the defaults; SCA_KD_WAVE_CAP picks a smaller workgroup form (sca_ctx::kd_wave_cap)
constexpr int KD_WAVE_FLOOR = 128;                     // tables are sized for subtrees handed over at this size or above
constexpr int KD_MAX_LEVELS = 40;
constexpr int KD_CHUNK = 2048;         // positions per workgroup in the level passes over larger nodes

// why a build reported failure (bits of counts[KD_MAX_LEVELS + 1]; sticky across builds until sca_synchronize reports them)
enum { KD_ERR_SPIN = 1, KD_ERR_CHUNKS = 4, KD_ERR_JOBS = 8, KD_ERR_LEVELS = 16, KD_ERR_BLOCK = 128 };
struct KdJob { int begin, end, node, pad; };
// workgroup -> node of a level pass: job index, first workgroup of the node, the node's extent and its split plane
// (axis < 0: not known when the record was written -- the root -- take it from the node's box)
struct alignas(16) KdChunkRec { int job, first, nb, ne, axis, pad; double split; };

struct KdScratch {
    double *kx, *ky, *kz;     // [n] coordinates in position order
    int *mr;                  // [n] mr[b + j - 1] = position of the j-th "< split" member of the node that starts at b
    KdJob *jobs[2];           // ping-pong lists of nodes with > wave_max members
    KdJob *small;             // [n] subtrees handed to k_kd_block
    int *counts;              // [KD_MAX_LEVELS + 2] jobs per level; [KD_MAX_LEVELS] = small count; [KD_MAX_LEVELS+1] = overflow flag;
                              // behind nchunks, [2 * KD_MAX_LEVELS + 3] = table slots handed out by the tail launch
    int job_cap;
    int wave_max;             // nodes up to this size go to k_kd_block (KD_WAVE_MIN < wave_max <= KD_WAVE_CAP)
    // multi-workgroup level passes
    unsigned long long *nbox; // [2][job_cap][6] order-preserving keys of the node boxes (per level parity)
    int *nge;                 // [2][job_cap] number of members >= split per node
    int *ps;                  // [n] inclusive count of ">= split" members inside the node up to the position
    int chunk_cap;
    unsigned long long *cbox; // [2][job_cap][2][6] boxes of the two children, accumulated while the parent is partitioned
    unsigned long long *chain;// [chunk_cap] chained scan: (launch token << 32) | number of ">= split" members of the chunk
    KdChunkRec *chunks[2];    // [chunk_cap] per level parity: everything a workgroup of a level pass needs, in one 32-byte read
    int *nchunks;             // [KD_MAX_LEVELS + 1] workgroups with work per level
    int *ticket;              // [KD_MAX_LEVELS + 1] arrival counters of k_kd_lv_rank<true> (chunk index by arrival, see there)
    int skip_prep;            // 1: k_kd_gather leaves the prologue of the tracker's agents to the tracker's kernels (v_pref is still being computed by
                              //    the tracker on another stream)
    int aux;                  // 1: the build runs BESIDE the pass that owns the step's counters and prologue (SCA_NBR_AUTO: the grid's kernels have them):
                              //    k_kd_gather touches neither
};

// SCA_NBR_AUTO (round 6): the kd query of the agents the pass's grid query listed WITHOUT a launch of its own (k_neighbors_kd_auto)
// and without a stream wait operation on the pass's chain.
// The kd build of an AUTO pass is a loop of its own beside the step -- every build starts from the previous one's permutation -- and at
// N = 4096 that loop and the pass are equally long chains of dependent dispatches (profiles/r06_a_c3_auto_device_timeline.json: the loop
// gather 3.0 + top 23.3 + block 32.0 + kd query 0.9 us of kernels, 19.7 us of gaps = the 78.5-us step; the pass 40.6 us of kernels in
// 73.0).  The query's launch cost the loop two of its four gaps, one of them a cross-stream event wait, to find its list empty; and the
// pass paid a hipStreamWaitValue32 -- a 2-us kernel of the runtime's and its gaps, 8.6 us -- in front of the solve for the same news.
// Now: k_kd_block's workgroups count out by ticket and the last one publishes "the tree of pass `seq` is complete" (one word); the grid
// query's workgroups count out by ticket and the last one reads the list's length.  Nobody listed (the usual case): done, the launch
// ends, the solve follows -- no wait, no launch, no fence.  Somebody listed: that workgroup waits for the tree's word and answers the
// listed agents on the spot, a wavefront per agent (kd_answer_listed), before the launch ends.  The wait is safe: a pass's build is
// ALWAYS enqueued before its grid query in the host's order (ahead, behind the previous step's integrate stage, or inside the pass in
// front of the grid build), so whether the two streams share a hardware queue (the build then ran first) or not (it runs beside, on
// other CUs: one spinning workgroup starves nobody) the word comes.  (The first version had it the other way round -- the BUILD's last
// workgroup waited for the grid query, a LATER launch -- and the suite that takes four minutes took fifteen: a kernel that waits for a
// launch behind it in a shared hardware queue waits for ever.)  The host takes this form while the list lengths that come back say
// "nobody" (sca_hip.hip: auto_tail_max = 0); a list that appears is answered here by this one workgroup for the few passes until the
// host has seen its length, then by the launch form with its 64 .. 1024 workgroups.
struct KdTail {
    unsigned seq;             // the pass this build belongs to (sca_ctx::auto_seq of its grid query); 0: publish nothing
    unsigned *sync;           // [0] k_kd_block's ticket, [1] the last pass whose tree is complete, [2] the grid query's ticket
};

// the kd query of the listed agents by ONE workgroup (every thread calls it); `stacks`: NW x KD_RSTACK x 16 doubles of LDS
template <int NW, bool HAS_OBS = true>
__device__ __forceinline__ void kd_answer_listed(const DeviceView &d, const Params &P, double agent_reach, double obs_reach, double max_radius,
                                                 int n, double (*stacks)[16]) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double (*rst)[16] = stacks + (size_t)wid * KD_RSTACK;
    if (n <= d.kdq_cap) {
        for (int i = wid; i < n; i += NW) neighbors_one<HAS_OBS>(d, P, agent_reach, obs_reach, max_radius, rst, d.kdq_list[i], lane);
    } else {                                                             // "too many for a list": every agent of the shard
        for (int i = wid; i < d.shard_count; i += NW) neighbors_one<HAS_OBS>(d, P, agent_reach, obs_reach, max_radius, rst, d.shard_begin + i, lane);
    }
}

// order-preserving map double -> u64 so that integer atomics give exact min / max
__device__ __forceinline__ unsigned long long dkey(double x) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dunkey(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k ^ 0x8000000000000000ull) : ~k;
    return __longlong_as_double((long long)u);
}

__device__ __forceinline__ double wave_min_d(double v) { return wave_min_f64(v); }
__device__ __forceinline__ double wave_max_d(double v) { return wave_max_f64(v); }
__device__ __forceinline__ int wave_sum_i(int v) { return wave_sum_i32(v); }

// Coordinates into position order, and the root's box (its accumulator was reset by the previous build's last kernel).
__global__ __launch_bounds__(256) void k_kd_gather(DeviceView d, KdScratch s, Params P) {
    SCA_TL(d, TL_KD_GATHER);
    SCA_KD_SETPRIO();
    __shared__ double red[4][6];
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (p == 0) {
        for (int i = 0; i < KD_MAX_LEVELS + 1; i++) s.counts[i] = 0;      // not the error word: it stays until the host has read it
        KdJob j; j.begin = 0; j.end = d.n; j.node = 0; j.pad = -1;          // pad = 2 * parent + side, -1 for the root
        if (d.n > s.wave_max) { s.jobs[0][0] = j; s.counts[0] = 1; }
        else { s.small[0] = j; s.counts[KD_MAX_LEVELS] = 1; }
        for (int i = 1; i <= KD_MAX_LEVELS; i++) s.nchunks[i] = 0;
        s.counts[2 * KD_MAX_LEVELS + 3] = 0;                                 // table slots handed out by the tail launch
        for (int i = 0; i <= KD_MAX_LEVELS; i++) s.ticket[i] = 0;
        s.nchunks[0] = d.n > s.wave_max ? (d.n + KD_CHUNK - 1) / KD_CHUNK : 0;
    }
    if (d.n > s.wave_max && p < (d.n + KD_CHUNK - 1) / KD_CHUNK) {                                    // the root's workgroups
        KdChunkRec r; r.job = 0; r.first = 0; r.nb = 0; r.ne = d.n; r.axis = -1; r.pad = 0; r.split = 0.0;
        s.chunks[0][p] = r;
    }
    if (!s.aux) {
        if (p < 256) d.done_count[p * 32] = 0;                               // start of a step: K4's counters
        if (p == 0) *d.fb_count = 0;                                         // ... and an empty fallback list
    }
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (p < d.n) {
        const int id = d.aperm[p];
        // the three position fields and nothing else of the record: a build beside a pass (aux, SCA_NBR_AUTO's build-ahead in
        // sca_run_steps) runs while k_collide_finish / k_goal_flags_others / the next k_action write `.flags` (and the velocity) of the same
        // records on the main stream -- the positions are final by then (ev_auto_moved), the rest is not this kernel's to look at
        static_assert(offsetof(PubRec, px) == 0 && offsetof(PubRec, py) == 8 && offsetof(PubRec, pz) == 16, "PubRec starts with the position");
        const double *rp = &d.rec[id].px;
        const double rx = rp[0], ry = rp[1], rz = rp[2];
        s.kx[p] = rx; s.ky[p] = ry; s.kz[p] = rz;
        mn[0] = mx[0] = rx; mn[1] = mx[1] = ry; mn[2] = mx[2] = rz;
        // every agent appears once in the permutation; the prologue is only needed for the rank's own shard
        if (!s.aux && shard_owns(d, id) && !(s.skip_prep && tracker_owns(d, id))) prep_agent<PREP_LIBM>(d, P, (Prep *)d.prep, id);
    }
    if (d.n > s.wave_max) {
#pragma unroll
        for (int k = 0; k < 3; k++) { mn[k] = wave_min_d(mn[k]); mx[k] = wave_max_d(mx[k]); }
        if (lane == 0) { for (int k = 0; k < 3; k++) { red[wid][k] = mn[k]; red[wid][3 + k] = mx[k]; } }
        __syncthreads();
        if (threadIdx.x < 6) {
            const int t = threadIdx.x;
            double v = red[0][t];
            for (int w = 1; w < 4; w++) v = t < 3 ? (red[w][t] < v ? red[w][t] : v) : (red[w][t] > v ? red[w][t] : v);
            if (t < 3) atomicMin(&s.nbox[t], dkey(v)); else atomicMax(&s.nbox[t], dkey(v));
        }
    }
}

// every node publishes its header in its own query record and its box in its parent's (KdWide)
__device__ __forceinline__ void kd_publish(KdWide *wide, const KdNode &nd, int node, int parent_code) {
    KdWide *w = &wide[node];
    w->begin = nd.begin; w->end = nd.end; w->left = nd.left; w->right = nd.right;
    if (parent_code >= 0) {
        KdWide *pw = &wide[parent_code >> 1];
        const int side = parent_code & 1;
        for (int k = 0; k < 3; k++) { pw->bx[kdw_idx(side, 0, k)] = nd.mn[k]; pw->bx[kdw_idx(side, 1, k)] = nd.mx[k]; }
    }
}

// kdTree.py:89-96: split axis and plane from the box
__device__ __forceinline__ void kd_split(const double mn[3], const double mx[3], int &axis, double &split) {
    const double d0 = mx[0] - mn[0], d1 = mx[1] - mn[1], d2 = mx[2] - mn[2];
    axis = (d0 > d1 && d0 > d2) ? 0 : (d1 > d2 ? 1 : 2);
    split = 0.5 * (mx[axis] + mn[axis]);
}

// ---- level passes over the large nodes, one workgroup per chunk of KD_CHUNK positions ---------------------------------
constexpr int KD_LV_T = 512;
constexpr int KD_LV_E = KD_CHUNK / KD_LV_T;     // 4 strided positions per thread (tile t covers [t*T, (t+1)*T))


// workgroup -> (node, chunk, split plane) through the level's chunk table (written by the parent level's swap pass)
struct KdChunk { int job, begin, end, node_begin, node_end, first_chunk, valid, axis; double split; };
__device__ __forceinline__ KdChunk kd_find_chunk(const KdScratch &s, int level, int blk) {
    KdChunk c; c.valid = 0; c.job = 0; c.begin = c.end = c.node_begin = c.node_end = c.first_chunk = 0; c.axis = 0; c.split = 0.0;
    const int nch = s.nchunks[level];
    const KdChunkRec rec = s.chunks[level & 1][blk];             // blk < chunk_cap always: read before the bound is known
    if (blk >= nch) return c;
    c.valid = 1; c.job = rec.job; c.node_begin = rec.nb; c.node_end = rec.ne; c.first_chunk = rec.first;
    c.begin = rec.nb + (blk - rec.first) * KD_CHUNK;
    c.end = c.begin + KD_CHUNK < rec.ne ? c.begin + KD_CHUNK : rec.ne;
    c.axis = rec.axis; c.split = rec.split;
    if (rec.axis < 0) {
        const unsigned long long *box = s.nbox + ((size_t)(level & 1) * s.job_cap + rec.job) * 6;
        double mn[3], mx[3];
        for (int k = 0; k < 3; k++) { mn[k] = dunkey(box[k]); mx[k] = dunkey(box[3 + k]); }
        kd_split(mn, mx, c.axis, c.split);
    }
    return c;
}

__device__ __forceinline__ void kd_node_split(const KdScratch &s, int level, int job, int &axis, double &split, double mn[3], double mx[3]) {
    const unsigned long long *box = s.nbox + ((size_t)(level & 1) * s.job_cap + job) * 6;
    for (int k = 0; k < 3; k++) { mn[k] = dunkey(box[k]); mx[k] = dunkey(box[3 + k]); }
    kd_split(mn, mx, axis, split);
}

// Level pass A: one workgroup per chunk.  The node's box is complete (accumulated by the parent's pass), so the chunk
// can flag its members against the split plane, scan the flags (chained across the chunks of the node through one
// 64-bit word per chunk: launch token | count), write the in-node ranks, and accumulate the boxes of the two children.
struct KdRankLds { int wtot[KD_LV_T / 64]; int carry_sh; double red[KD_LV_T / 64][12]; int wt[KD_CHUNK / KD_LV_T][KD_LV_T / 64]; };
__device__ __forceinline__ void kd_rank_part(const KdScratch s, int level, unsigned token, const KdChunk c, KdRankLds &SH, int blk) {
    constexpr int W = KD_LV_T / 64;
    int (&wtot)[W] = SH.wtot;
    int &carry_sh = SH.carry_sh;
    double (&red)[W][12] = SH.red;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int axis = c.axis;
    const double split = c.split;
    const int b = c.node_begin;
    // ---- flags of this thread's positions (tile t covers [begin + t*T, begin + (t+1)*T)), chunk total
    unsigned flags = 0;
    int mine = 0;
    double cmn[2][3] = {{INFINITY, INFINITY, INFINITY}, {INFINITY, INFINITY, INFINITY}};
    double cmx[2][3] = {{-INFINITY, -INFINITY, -INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
#pragma unroll
    for (int t = 0; t < KD_LV_E; t++) {
        const int p = c.begin + t * KD_LV_T + tid;
        if (p < c.end) {
            const double x = s.kx[p], y = s.ky[p], z = s.kz[p];
            const double cv = axis == 0 ? x : (axis == 1 ? y : z);
            const int side = cv < split ? 0 : 1;
            if (side) { flags |= 1u << t; mine++; }
            cmn[side][0] = x < cmn[side][0] ? x : cmn[side][0]; cmx[side][0] = x > cmx[side][0] ? x : cmx[side][0];
            cmn[side][1] = y < cmn[side][1] ? y : cmn[side][1]; cmx[side][1] = y > cmx[side][1] ? y : cmx[side][1];
            cmn[side][2] = z < cmn[side][2] ? z : cmn[side][2]; cmx[side][2] = z > cmx[side][2] ? z : cmx[side][2];
        }
    }
    const int wsum = wave_sum_i(mine);
    if (lane == 0) wtot[wid] = wsum;
    // the ">= split" lanes of every tile, and the wavefront's count per tile for the ranks below (until round 4 the rank loop
    // exchanged them tile by tile: eight workgroup barriers per chunk)
    unsigned long long tm[KD_LV_E];
#pragma unroll
    for (int t = 0; t < KD_LV_E; t++) {
        tm[t] = __ballot((flags >> t) & 1u);
        if (lane == 0) SH.wt[t][wid] = __popcll(tm[t]);
    }
    // children boxes: wave reduce, then one set of atomics per workgroup
#pragma unroll
    for (int sd = 0; sd < 2; sd++)
#pragma unroll
        for (int k = 0; k < 3; k++) { cmn[sd][k] = wave_min_d(cmn[sd][k]); cmx[sd][k] = wave_max_d(cmx[sd][k]); }
    if (lane == 0)
        for (int sd = 0; sd < 2; sd++)
            for (int k = 0; k < 3; k++) { red[wid][sd * 6 + k] = cmn[sd][k]; red[wid][sd * 6 + 3 + k] = cmx[sd][k]; }
    __syncthreads();
    int total = 0;
    for (int w = 0; w < W; w++) total += wtot[w];
    if (tid < 12) {
        const bool is_min = (tid % 6) < 3;
        double v = red[0][tid];
        for (int w = 1; w < W; w++) v = is_min ? (red[w][tid] < v ? red[w][tid] : v) : (red[w][tid] > v ? red[w][tid] : v);
        unsigned long long *cb = s.cbox + ((size_t)(level & 1) * s.job_cap + c.job) * 12;
        if (is_min) atomicMin(&cb[tid], dkey(v)); else atomicMax(&cb[tid], dkey(v));
    }
    // ---- chained scan: publish this chunk's count, collect the predecessors' (all chunks of a level are co-resident).
    //      The first wavefront reads 64 predecessors at a time: the words are independent, only their arrival is awaited.
    if (tid == 0)
        __hip_atomic_store(&s.chain[blk], ((unsigned long long)token << 32) | (unsigned)total, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    if (wid == 0) {
        int carry = 0;
        for (int ch = c.first_chunk + lane; ch < blk; ch += 64) {
            unsigned long long v = 0;
            int spins = 0;
            for (;;) {
                v = __hip_atomic_load(&s.chain[ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(v >> 32) == token) break;
                if (++spins > (1 << 24)) { atomicOr(&s.counts[KD_MAX_LEVELS + 1], KD_ERR_SPIN); break; }     // never hang the GPU: report
                __builtin_amdgcn_s_sleep(2);
            }
            carry += (int)(unsigned)v;
        }
        carry = wave_sum_i(carry);
        if (lane == 0) {
            carry_sh = carry;
            const int nch = (c.node_end - b + KD_CHUNK - 1) / KD_CHUNK;
            if (blk == c.first_chunk + nch - 1) s.nge[(level & 1) * s.job_cap + c.job] = carry + total;
        }
    }
    __syncthreads();
    int carry = carry_sh;
    // ---- ranks: ps = inclusive ">= split" count inside the node; mr[b + j - 1] = position of the j-th "< split" member
#pragma unroll
    for (int t = 0; t < KD_LV_E; t++) {
        const int p = c.begin + t * KD_LV_T + tid;
        const bool ge = (flags >> t) & 1u;
        const int incl_w = __popcll(tm[t] & ((2ull << lane) - 1ull));
        int woff = 0, ttot = 0;
#pragma unroll
        for (int w = 0; w < W; w++) { const int v = SH.wt[t][w]; if (w < wid) woff += v; ttot += v; }
        const int G = carry + woff + incl_w;
        if (p < c.end) {
            s.ps[p] = G;
            if (!ge) s.mr[b + ((p - b + 1) - G) - 1] = p;
        }
        carry += ttot;
    }
}

// TICKET: which chunk a workgroup takes.  The chained scan makes a chunk wait for the chunks in front of it in its node.  With
// chunk = blockIdx that is free of deadlock as long as the dispatcher starts workgroups in index order (per XCD, which is how
// it behaves: the lowest unfinished chunk is then always running) -- and certainly while every chunk of a level is resident
// at once.  A level with more chunks than the chip can hold at once (beyond ~2 million agents per rank) does not lean on that:
// its workgroups take their chunk by ARRIVAL (one atomic per workgroup), so a chunk can only ever wait for workgroups that
// already run.  The host picks the form per build from the occupancy the runtime reports (sca_hip.hip, kd_rank_capacity).
template <bool TICKET>
__global__ __launch_bounds__(KD_LV_T) void k_kd_lv_rank(KdScratch s, int level, unsigned token) {
    SCA_KD_SETPRIO();
    __shared__ KdRankLds SH;
    int blk = (int)blockIdx.x;
    if (TICKET) {
        __shared__ int tk;
        if (threadIdx.x == 0) tk = atomicAdd(&s.ticket[level], 1);
        __syncthreads();
        blk = tk;
    }
    const KdChunk c = kd_find_chunk(s, level, blk);
    if (!c.valid) return;
    kd_rank_part(s, level, token, c, SH, blk);
}

// Level pass B: swaps (kdTree.py:108-111), node record and children (kdTree.py:112-122)
// the children that need another level pass, as the chunk records written for them (for k_kd_level_tail)
struct KdTailOut { int n; int slot_base; KdChunkRec r[2]; };   // slot_base: first free table slot of the tail launch
__device__ __forceinline__ void kd_swap_part(const DeviceView d, const KdScratch s, int level, const KdChunk c, KdTailOut *tail = nullptr) {
    const int tid = threadIdx.x;
    const KdJob *in = s.jobs[level & 1];
    KdJob *out = s.jobs[(level + 1) & 1];
    const int axis = c.axis;
    const double split = c.split;
    const double *kc = axis == 0 ? s.kx : (axis == 1 ? s.ky : s.kz);
    const int b = c.node_begin, e = c.node_end;
    const int L = (e - b) - s.nge[(level & 1) * s.job_cap + c.job];
    const int lim = c.end < b + L ? c.end : b + L;
    // (Taking a thread's positions two or four at a time -- flags, partners, both records, stores: four trips to memory per chunk
    // instead of four per position -- was measured in round 4: slower.  Two at a time 11.2 us against 10.0 at c5, 7.9 against 6.6
    // at c4: the bookkeeping wavefront below sets the length, not this loop.  Four at a time needs 121 registers, and then a
    // workgroup of nine wavefronts no longer finds room on the CUs whose SIMDs hold a 256-register re-plan wavefront each:
    // 34 us at c5, 45 at c4.  The build's kernels stay under 256 / 3 registers for that reason.)
    for (int p = c.begin + tid; p < lim && tid < KD_LV_T; p += KD_LV_T) {
        if (!(kc[p] < split)) {
            // the k-th ">= split" member of the left part (k = ps - 1) takes the k-th "< split" member from the right,
            // i.e. the (L - k)-th "< split" member of the node
            const int q = s.mr[b + (L - (s.ps[p] - 1)) - 1];
            const int ip = d.aperm[p], iq = d.aperm[q];
            d.aperm[p] = iq; d.aperm[q] = ip;
            const double xp = s.kx[p], yp = s.ky[p], zp = s.kz[p];
            s.kx[p] = s.kx[q]; s.ky[p] = s.ky[q]; s.kz[p] = s.kz[q];
            s.kx[q] = xp; s.ky[q] = yp; s.kz[q] = zp;
        }
    }
    // The node's bookkeeping (its record, the children and their chunk records) is a chain of dependent loads and atomics
    // with returns, ~5 us long and independent of the swaps: k_kd_lv_swap gives it a wavefront of its own (thread KD_LV_T of
    // a KD_LV_T + 64 workgroup) so that it runs beside the swaps, not behind thread 0's; the tail launch keeps thread 0.
    // Everything it reads was completed by the rank pass; all loads and all four atomics are issued before anything waits.
    const int keeper = blockDim.x > KD_LV_T ? KD_LV_T : 0;
    if (tid == keeper && c.begin == b) {
        if (tail) tail->n = 0;
        const KdJob job = in[c.job];
        const unsigned long long *nb_ = s.nbox + ((size_t)(level & 1) * s.job_cap + c.job) * 6;
        const unsigned long long *cb_ = s.cbox + ((size_t)(level & 1) * s.job_cap + c.job) * 12;
        unsigned long long bk[6], cb[12];
#pragma unroll
        for (int q = 0; q < 6; q++) bk[q] = nb_[q];
#pragma unroll
        for (int q = 0; q < 12; q++) cb[q] = cb_[q];
        const int leftSize = L == 0 ? 1 : L;                 // degenerate: every member on the split plane
        const int cbeg[2] = {b, b + leftSize}, cend[2] = {b + leftSize, e};
        bool big[2]; int nch[2], arrival[2] = {0, 0}, base[2] = {0, 0}, slot[2] = {0, 0}, small_at[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < 2; k++) {
            big[k] = cend[k] - cbeg[k] > s.wave_max;
            nch[k] = (cend[k] - cbeg[k] + KD_CHUNK - 1) / KD_CHUNK;
            if (big[k]) {
                if (level + 1 < KD_MAX_LEVELS) {
                    // table slot of the child.  A level launch owns the level's slots: index = arrival order.  In the tail
                    // launch workgroups are at DIFFERENT levels at the same time, so slots (and chunk records) indexed per
                    // level parity would collide: there the index comes from one counter that starts behind the tail
                    // level's own nodes (unique over both parities), the chunk records stay in the workgroup's stack, and
                    // the per-level counters are only statistics for the host.
                    arrival[k] = atomicAdd(&s.counts[level + 1], 1);
                    base[k] = atomicAdd(&s.nchunks[level + 1], nch[k]);
                    if (tail) slot[k] = atomicAdd(&s.counts[2 * KD_MAX_LEVELS + 3], 1);
                }
            } else small_at[k] = atomicAdd(&s.counts[KD_MAX_LEVELS], 1);
        }
        double mn[3], mx[3];
        for (int k = 0; k < 3; k++) { mn[k] = dunkey(bk[k]); mx[k] = dunkey(bk[3 + k]); }
        KdNode nd;
        nd.begin = b; nd.end = e; nd.left = job.node + 1; nd.right = job.node + 2 * leftSize;
        for (int k = 0; k < 3; k++) { nd.mn[k] = mn[k]; nd.mx[k] = mx[k]; }
        if (job.node == 0) d.atree[0] = nd;             // the root's box has no parent record to live in
        kd_publish(d.awide, nd, job.node, job.pad);
        if (L == 0) {
            // nobody below the midpoint: the box has no extent, every member sits on one point (kdTree.py:113-116 then puts
            // one of them left, the rest right) -- both children have the parent's box
            for (int q = 0; q < 3; q++) { cb[q] = cb[6 + q] = dkey(mn[q]); cb[3 + q] = cb[9 + q] = dkey(mx[q]); }
        }
        KdJob ch[2];
        ch[0].begin = cbeg[0]; ch[0].end = cend[0]; ch[0].node = nd.left; ch[0].pad = 2 * job.node;
        ch[1].begin = cbeg[1]; ch[1].end = cend[1]; ch[1].node = nd.right; ch[1].pad = 2 * job.node + 1;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (big[k]) {
                if (level + 1 < KD_MAX_LEVELS) {
                    const int at = tail ? tail->slot_base + slot[k] : arrival[k];
                    if (at < s.job_cap) {
                        out[at] = ch[k];
                        unsigned long long *box = s.nbox + ((size_t)((level + 1) & 1) * s.job_cap + at) * 6;
                        for (int q = 0; q < 6; q++) box[q] = cb[k * 6 + q];          // the child's box is already known
                        unsigned long long *ncb = s.cbox + ((size_t)((level + 1) & 1) * s.job_cap + at) * 12;
                        for (int q = 0; q < 12; q++) ncb[q] = (q % 6) < 3 ? dkey(INFINITY) : dkey(-INFINITY);
                        // the child's workgroups of the next level
                        if (tail || base[k] + nch[k] <= s.chunk_cap) {
                            KdChunkRec r; r.job = at; r.first = base[k]; r.nb = ch[k].begin; r.ne = ch[k].end; r.pad = 0;
                            double cmn[3], cmx[3];
                            for (int q = 0; q < 3; q++) { cmn[q] = dunkey(cb[k * 6 + q]); cmx[q] = dunkey(cb[k * 6 + 3 + q]); }
                            kd_split(cmn, cmx, r.axis, r.split);
                            if (tail) tail->r[tail->n++] = r;
                            else {
                                KdChunkRec *tab = s.chunks[(level + 1) & 1];
                                for (int q = 0; q < nch[k]; q++) tab[base[k] + q] = r;
                            }
                        } else atomicOr(&s.counts[KD_MAX_LEVELS + 1], KD_ERR_CHUNKS);
                    } else atomicOr(&s.counts[KD_MAX_LEVELS + 1], KD_ERR_JOBS);
                } else atomicOr(&s.counts[KD_MAX_LEVELS + 1], KD_ERR_LEVELS);
            } else s.small[small_at[k]] = ch[k];
        }
    }
}

__global__ __launch_bounds__(KD_LV_T + 64) void k_kd_lv_swap(DeviceView d, KdScratch s, int level) {
    SCA_TL(d, TL_KD_LEVELS);
    SCA_KD_SETPRIO();
    const KdChunk c = kd_find_chunk(s, level, blockIdx.x);
    if (!c.valid) return;
    kd_swap_part(d, s, level, c);
}

// The LAST level launch of a build: one workgroup per node of that level finishes everything that is left below it -- both
// passes of the node (its chunks one after the other: ranks, `__syncthreads`, swaps), then the same for every child that
// still has more than wave_max members, depth first from a small stack.  At the level the host picks (from the previous
// builds' statistics: the first one whose nodes are about one chunk) that is one short pass per workgroup; if the tree has
// changed since -- a node with many chunks, more levels -- it is slower, never wrong: the statistics only set the speed.
constexpr int KD_TAIL_STACK = 2 * KD_MAX_LEVELS;
__global__ __launch_bounds__(KD_LV_T) void k_kd_level_tail(DeviceView d, KdScratch s, int level, unsigned token) {
    SCA_TL(d, TL_KD_LEVELS);
    SCA_KD_SETPRIO();
    __shared__ KdRankLds SH;
    __shared__ KdTailOut out;
    __shared__ KdChunkRec stack[KD_TAIL_STACK];
    __shared__ int stack_lv[KD_TAIL_STACK];
    __shared__ int sp_sh;
    __shared__ KdChunkRec cur_rec;
    __shared__ int cur_lv;
    const int blk = (int)blockIdx.x;
    if (blk >= s.nchunks[level]) return;
    const KdChunk c0 = kd_find_chunk(s, level, blk);
    if (c0.first_chunk != blk) return;                              // one workgroup per node: the one of its first chunk
    if (threadIdx.x == 0) {
        KdChunkRec r; r.job = c0.job; r.first = c0.first_chunk; r.nb = c0.node_begin; r.ne = c0.node_end; r.axis = c0.axis; r.pad = 0; r.split = c0.split;
        stack[0] = r; stack_lv[0] = level; sp_sh = 1;
        out.slot_base = s.counts[level];                           // behind the nodes this launch starts from
    }
    const int chain0 = c0.first_chunk;                              // this workgroup's own range of chain words
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            if (sp_sh > 0) { sp_sh--; cur_rec = stack[sp_sh]; cur_lv = stack_lv[sp_sh]; }
            else cur_lv = -1;
        }
        __syncthreads();
        const int lv = cur_lv;
        if (lv < 0) break;
        const KdChunkRec r = cur_rec;
        const int nch = (r.ne - r.nb + KD_CHUNK - 1) / KD_CHUNK;
        KdChunk c; c.valid = 1; c.job = r.job; c.node_begin = r.nb; c.node_end = r.ne; c.first_chunk = chain0; c.axis = r.axis; c.split = r.split;
        for (int t = 0; t < nch; t++) {
            c.begin = r.nb + t * KD_CHUNK;
            c.end = c.begin + KD_CHUNK < r.ne ? c.begin + KD_CHUNK : r.ne;
            kd_rank_part(s, lv, token, c, SH, chain0 + t);
            __syncthreads();
        }
        for (int t = 0; t < nch; t++) {
            c.begin = r.nb + t * KD_CHUNK;
            c.end = c.begin + KD_CHUNK < r.ne ? c.begin + KD_CHUNK : r.ne;
            kd_swap_part(d, s, lv, c, &out);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 0; k < out.n; k++) {
                if (sp_sh < KD_TAIL_STACK) { stack[sp_sh] = out.r[k]; stack_lv[sp_sh] = lv + 1; sp_sh++; }
                else atomicOr(&s.counts[KD_MAX_LEVELS + 1], KD_ERR_LEVELS);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// One workgroup finishes one subtree of <= KBM members in LDS.  Every level is one element-parallel pass over all of the
// subtree's positions: each thread owns two consecutive positions.  Two instantiations: 1024 members / 512 threads (69 KB
// of LDS: two workgroups per CU) and 1536 / 768 (one per CU), picked by the build's wave_max.
//
// x, y, z and slot are read both as 2 and as 8 consecutive positions per lane; position p lives at (p & 7) * (KBM / 8 + 8)
// + (p >> 3), so that both patterns touch all LDS banks evenly (rows of KBM / 8 + 8: four rows of doubles tile the 64 banks)
#define KB_SW(p) ((((p) & 7) * KB_SWROW) + ((p) >> 3))

template <int KBM, int KBT>
struct KbLds {
    static constexpr int SWROW = KBM / 8 + 8, SWN = 8 * SWROW, NODES = (2 * KBM) / 11 + 8;   // live nodes per level <= 2 * KBM / 11
    double x[SWN], y[SWN], z[SWN];          // swizzled: index KB_SW(position)
    int id[KBM];
    int slot[SWN];                          // live-node slot of the position, -1 once its leaf is written
    int mr[KBM];                            // position of the k-th misplaced member of the right part
    int ps[KBM];                            // inclusive prefix count of the ">= split" flags over all positions
    int nb[2][NODES], ne[2][NODES], nnode[2][NODES], npar[2][NODES];
    unsigned long long box[2][NODES][6];
    int child[NODES], lfix[NODES];
    int count[2];
    int wtot[KBT / 64];
};

// Barrier for LDS-only hand-offs inside k_kd_block's level loop: __syncthreads() also waits for the global stores of the
// node records (s_waitcnt vmcnt(0)), a ~1 us round trip per level that nothing in the loop depends on.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// minimum of a 64-bit key over the 16-lane row (every lane of the row gets it) or over the wavefront (uniform result):
// two chains of 32-bit DPP minima, high words first, then the low words of the lanes that hold the minimal high word
template <bool WAVE>
__device__ __forceinline__ unsigned long long kb_key_min(unsigned long long key) {
    const unsigned hi = (unsigned)(key >> 32), lo = (unsigned)key;
    auto red = [](unsigned v) {
        v = umin32(v, (unsigned)dpp_mov<0xb1, 0xf>((int)v));
        v = umin32(v, (unsigned)dpp_mov<0x4e, 0xf>((int)v));
        v = umin32(v, (unsigned)dpp_mov<0x124, 0xf>((int)v));
        v = umin32(v, (unsigned)dpp_mov<0x128, 0xf>((int)v));
        if (WAVE) {
            v = umin32(v, (unsigned)dpp_mov<0x142, 0xa>((int)v));
            v = umin32(v, (unsigned)dpp_mov<0x143, 0xc>((int)v));
            v = (unsigned)__builtin_amdgcn_readlane((int)v, 63);
        }
        return v;
    };
    const unsigned h = red(hi);
    const unsigned l = red(hi == h ? lo : 0xffffffffu);
    return ((unsigned long long)h << 32) | l;
}
// inclusive prefix sum over the wavefront: shifts inside the rows, then the row totals carried across
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
    const int lane = (int)(threadIdx.x & 63);
    int t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); v += t;      // row_shr:1 (lanes without a source add 0)
    t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); v += t;      // row_shr:2
    t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); v += t;      // row_shr:4
    t = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); v += t;      // row_shr:8
    const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47);
    const int row = lane >> 4;
    return v + (row > 0 ? r0 : 0) + (row > 1 ? r1 : 0) + (row > 2 ? r2 : 0);
}

#ifdef SCA_KB_TIMING   // per-phase wall-clock ticks of workgroup 0's first job into s.ps (debug builds only)
#define KB_MARK_INIT() int dbg_i = 0; long long dbg_t = wall_clock64()
#define KB_MARK() do { if (jb == 0 && blockIdx.x == 0 && tid == 0 && dbg_i < 200) { const long long t_ = wall_clock64(); s.ps[dbg_i++] = (int)(t_ - dbg_t); dbg_t = t_; } } while (0)
#else
#define KB_MARK_INIT() do { } while (0)
#define KB_MARK() do { } while (0)
#endif

// KdTail: see there.  Every workgroup of k_kd_block comes through here when its subtrees are done; the last one says so.
__device__ __forceinline__ void kd_publish_tree(const KdTail &T) {
    __syncthreads();                                                     // (the workgroup's last subtree is written)
    if (threadIdx.x == 0) {
        __threadfence();                                                 // ... and visible before the ticket says so
        if (atomicAdd(T.sync, 1u) == gridDim.x - 1) {                    // the build's last workgroup: the tree is complete
            atomicExch(T.sync, 0u);                                      // (the next build's ticket starts at 0)
            __hip_atomic_store(T.sync + 1, T.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int KBM, int KBT>
__global__ __launch_bounds__(KBT) void k_kd_block(DeviceView d, KdScratch s, int levels_run, KdTail T) {
    SCA_TL(d, TL_KD_BLOCK);
    SCA_KD_SETPRIO();
    static_assert(KBM == 2 * KBT, "k_kd_block is written for two consecutive positions per thread");
    constexpr int KB_SWROW = KbLds<KBM, KBT>::SWROW, KB_MAX = KBM, KB_T = KBT, KB_E = 2;
    __shared__ KbLds<KBM, KBT> S;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int njobs = s.counts[KD_MAX_LEVELS];
    (void)levels_run;                      // k_kd_level_tail leaves no node larger than wave_max behind
    // the level passes are over: reset the root's accumulators (box, children boxes) for the next build
    if (blockIdx.x == 0 && tid < 12) {
        if (tid < 6) s.nbox[tid] = tid < 3 ? dkey(INFINITY) : dkey(-INFINITY);
        s.cbox[tid] = (tid % 6) < 3 ? dkey(INFINITY) : dkey(-INFINITY);
    }
    for (int jb = blockIdx.x; jb < njobs; jb += gridDim.x) {
        const KdJob job = s.small[jb];
        const int base = job.begin, size = job.end - job.begin;
        if (size > KB_MAX) { if (tid == 0) atomicOr(&s.counts[KD_MAX_LEVELS + 1], KD_ERR_BLOCK); continue; }
        __syncthreads();
        for (int i = tid; i < size; i += KB_T) {
            S.x[KB_SW(i)] = s.kx[base + i]; S.y[KB_SW(i)] = s.ky[base + i]; S.z[KB_SW(i)] = s.kz[base + i]; S.id[i] = d.aperm[base + i];
            S.slot[KB_SW(i)] = 0;
        }
        if (tid == 0) {
            S.nb[0][0] = 0; S.ne[0][0] = size; S.nnode[0][0] = job.node; S.npar[0][0] = job.pad; S.count[0] = 1; S.count[1] = 0;
            for (int k = 0; k < 3; k++) { S.box[0][0][k] = dkey(INFINITY); S.box[0][0][3 + k] = dkey(-INFINITY); }
        }
        __syncthreads();
        int cur = 0;
        bool first = true;
        const int p0 = tid * KB_E;
        KB_MARK_INIT();
        for (;;) {
            const int nc = S.count[cur];
            if (nc == 0) break;
            const int nxt = cur ^ 1;
            // ---- A: positions move to their child's slot (the previous level's result), then the boxes of the level's
            //         nodes (kdTree.py:63-83).  Two consecutive positions per lane.  All six extremes are carried as
            //         order-preserving 64-bit keys (maxima inverted, so every one is a minimum) and reduced as two 32-bit
            //         DPP min chains (high words, then the low words of the lanes that hold the minimum high word):
            //         over the whole wavefront when its 128 positions sit in one node, else over each 16-lane row whose
            //         32 positions do; only what is left goes through LDS atomics.
            const bool v0 = p0 < size, v1 = p0 + 1 < size;
            int s0 = v0 ? S.slot[KB_SW(p0)] : -1, s1 = v1 ? S.slot[KB_SW(p0 + 1)] : -1;
            if (!first) {
                int onew = -1, othr = 0;
                if (s0 >= 0) {
                    const int ob = S.nb[nxt][s0], oe = S.ne[nxt][s0];               // the parent level's node
                    if (oe - ob > MAX_LEAF) { onew = S.child[s0]; othr = ob + S.lfix[s0]; }
                }
                int n1 = -1;
                if (s1 >= 0) {
                    if (s1 == s0) n1 = onew < 0 ? -1 : onew + (p0 + 1 >= othr ? 1 : 0);
                    else {
                        const int ob = S.nb[nxt][s1], oe = S.ne[nxt][s1];
                        if (oe - ob > MAX_LEAF) n1 = S.child[s1] + (p0 + 1 >= ob + S.lfix[s1] ? 1 : 0);
                    }
                }
                s0 = (s0 >= 0 && onew >= 0) ? onew + (p0 >= othr ? 1 : 0) : -1;
                s1 = n1;
                if (v0) S.slot[KB_SW(p0)] = s0;
                if (v1) S.slot[KB_SW(p0 + 1)] = s1;
            }
            // extents of the positions' nodes
            int nb_[2] = {0, 0}, ne_[2] = {0, 0};
            if (s0 >= 0) { nb_[0] = S.nb[cur][s0]; ne_[0] = S.ne[cur][s0]; }
            if (s1 >= 0) {
                if (s1 == s0) { nb_[1] = nb_[0]; ne_[1] = ne_[0]; }
                else { nb_[1] = S.nb[cur][s1]; ne_[1] = S.ne[cur][s1]; }
            }
            const bool big0 = s0 >= 0, big1 = s1 >= 0;
            if (__any(big0 || big1)) {
                const double x0 = S.x[KB_SW(p0)], y0 = S.y[KB_SW(p0)], z0 = S.z[KB_SW(p0)];
                const int p1 = v1 ? p0 + 1 : p0;
                const double x1 = S.x[KB_SW(p1)], y1 = S.y[KB_SW(p1)], z1 = S.z[KB_SW(p1)];
                const bool pair = big0 && s1 == s0;
                unsigned long long key[6];                       // 0..2 minima, 3..5 inverted maxima of the lane's first segment
                {
                    const double a0 = pair && x1 < x0 ? x1 : x0, a1 = pair && y1 < y0 ? y1 : y0, a2 = pair && z1 < z0 ? z1 : z0;
                    const double b0 = pair && x1 > x0 ? x1 : x0, b1 = pair && y1 > y0 ? y1 : y0, b2 = pair && z1 > z0 ? z1 : z0;
                    key[0] = dkey(a0); key[1] = dkey(a1); key[2] = dkey(a2);
                    key[3] = ~dkey(b0); key[4] = ~dkey(b1); key[5] = ~dkey(b2);
                }
                // the second position on its own when it starts another node
                if (big1 && s1 != s0) {
                    atomicMin(&S.box[cur][s1][0], dkey(x1)); atomicMax(&S.box[cur][s1][3], dkey(x1));
                    atomicMin(&S.box[cur][s1][1], dkey(y1)); atomicMax(&S.box[cur][s1][4], dkey(y1));
                    atomicMin(&S.box[cur][s1][2], dkey(z1)); atomicMax(&S.box[cur][s1][5], dkey(z1));
                }
                const int rs = row_bcast_i<0>(s0);
                const unsigned long long um = __ballot(s0 == rs && (s1 == rs || !v1));
                const int ref = __builtin_amdgcn_readfirstlane(s0);
                const bool wave_uni = um == ~0ull && __all(rs == ref);
                if (wave_uni) {
                    if (ref >= 0) {
#pragma unroll
                        for (int k = 0; k < 6; k++) key[k] = kb_key_min<true>(key[k]);
                        if (lane == 0)
                            for (int k = 0; k < 3; k++) { atomicMin(&S.box[cur][ref][k], key[k]); atomicMax(&S.box[cur][ref][3 + k], ~key[3 + k]); }
                    }
                } else {
                    // segmented minimum scan along the row: a node's positions are consecutive, so equal slots at distance D
                    // mean one node in between; the last lane of every run holds the run's extremes and alone goes to LDS
                    unsigned long long rk[6];
                    for (int k = 0; k < 6; k++) rk[k] = key[k];
                    const int gl16 = lane & 15;
#define KB_SEG_STEP(D)                                                                                            \
                    {                                                                                             \
                        /* the DPP read runs for EVERY lane (a source lane masked off by a short-circuit && would read 0) */ \
                        const int ps_ = __builtin_amdgcn_mov_dpp(s0, 0x110 + D, 0xf, 0xf, true);                  \
                        const bool take = (gl16 >= D) & (ps_ == s0);                                              \
                        _Pragma("unroll") for (int k = 0; k < 6; k++) {                                           \
                            const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)rk[k], 0x110 + D, 0xf, 0xf, true);            \
                            const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(rk[k] >> 32), 0x110 + D, 0xf, 0xf, true);    \
                            const unsigned long long o = ((unsigned long long)hi << 32) | lo;                     \
                            if (take && o < rk[k]) rk[k] = o;                                                     \
                        }                                                                                         \
                    }
                    KB_SEG_STEP(1) KB_SEG_STEP(2) KB_SEG_STEP(4) KB_SEG_STEP(8)
#undef KB_SEG_STEP
                    const int nxt_slot = __builtin_amdgcn_mov_dpp(s0, 0x101, 0xf, 0xf, true);      // row_shl:1
                    const bool run_end = gl16 == 15 || nxt_slot != s0;
                    if (big0 && run_end) {
                        for (int k = 0; k < 3; k++) { atomicMin(&S.box[cur][s0][k], rk[k]); atomicMax(&S.box[cur][s0][3 + k], ~rk[3 + k]); }
                    }
                }
            }
            lds_barrier();
            KB_MARK();
            // ---- B: split plane of the element's node (kdTree.py:85-96), ">= split" flags, block scan
            bool live[2] = {false, false}, ge[2] = {false, false};
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int sl = k == 0 ? s0 : s1;
                if (sl >= 0) {
                    if (ne_[k] - nb_[k] > MAX_LEAF) {
                        live[k] = true;
                        double mn[3], mx[3];
                        for (int q = 0; q < 3; q++) { mn[q] = dunkey(S.box[cur][sl][q]); mx[q] = dunkey(S.box[cur][sl][3 + q]); }
                        int axis; double split;
                        kd_split(mn, mx, axis, split);
                        const int p = p0 + k;
                        const double c = axis == 0 ? S.x[KB_SW(p)] : (axis == 1 ? S.y[KB_SW(p)] : S.z[KB_SW(p)]);
                        ge[k] = !(c < split);
                    }
                }
            }
            const int tsum = (ge[0] ? 1 : 0) + (ge[1] ? 1 : 0);
            const int incl = wave_incl_scan_i32(tsum);
            if (lane == 63) S.wtot[wid] = incl;
            lds_barrier();
            KB_MARK();
            {
                // exclusive prefix of the wavefront totals: the 16 totals sit in one row, scanned with row shifts
                const int wv = lane < KB_T / 64 ? S.wtot[lane] : 0;
                const int wscan = wave_incl_scan_i32(wv);
                const int wbase = wid > 0 ? __builtin_amdgcn_readlane(wscan, wid - 1) : 0;
                const int excl = wbase + incl - tsum;
                if (v0) S.ps[p0] = excl + (ge[0] ? 1 : 0);
                if (v1) S.ps[p0 + 1] = excl + tsum;
            }
            lds_barrier();
            KB_MARK();
            // ---- C: L = #(members < split); the k-th "< split" member of the right part counted from the right
            int L_[2] = {0, 0}, G_[2] = {0, 0};
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (live[k]) {
                    const int p = p0 + k, b = nb_[k], e = ne_[k];
                    const int pb = b > 0 ? S.ps[b - 1] : 0;
                    L_[k] = (e - b) - (S.ps[e - 1] - pb);
                    G_[k] = S.ps[p] - pb;
                    if (!ge[k] && p >= b + L_[k]) S.mr[b + (L_[k] - ((p - b + 1) - G_[k]))] = p;
                }
            }
            lds_barrier();
            KB_MARK();
            // ---- D: the swaps (kdTree.py:108-111) by position ...
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int p = p0 + k;
                if (live[k] && ge[k] && p < nb_[k] + L_[k]) {
                    const int q = S.mr[nb_[k] + G_[k] - 1];
                    const int ip = S.id[p]; S.id[p] = S.id[q]; S.id[q] = ip;
                    double t;
                    t = S.x[KB_SW(p)]; S.x[KB_SW(p)] = S.x[KB_SW(q)]; S.x[KB_SW(q)] = t;
                    t = S.y[KB_SW(p)]; S.y[KB_SW(p)] = S.y[KB_SW(q)]; S.y[KB_SW(q)] = t;
                    t = S.z[KB_SW(p)]; S.z[KB_SW(p)] = S.z[KB_SW(q)]; S.z[KB_SW(q)] = t;
                }
            }
            //         ... and the node records and children (kdTree.py:112-122), one lane per node of the level
            if (tid < nc) {
                const int sl = tid;
                const int b = S.nb[cur][sl], e = S.ne[cur][sl];
                const int node = S.nnode[cur][sl], par = S.npar[cur][sl];
                KdNode nd;
                nd.begin = base + b; nd.end = base + e; nd.left = 0; nd.right = 0;
                for (int q = 0; q < 3; q++) { nd.mn[q] = dunkey(S.box[cur][sl][q]); nd.mx[q] = dunkey(S.box[cur][sl][3 + q]); }
                if (e - b > MAX_LEAF) {
                    const int pb = b > 0 ? S.ps[b - 1] : 0;
                    const int L = (e - b) - (S.ps[e - 1] - pb);
                    const int lf = L == 0 ? 1 : L;
                    nd.left = node + 1; nd.right = node + 2 * lf;
                    const int c0 = atomicAdd(&S.count[nxt], 2);
                    S.child[sl] = c0; S.lfix[sl] = lf;
                    S.nb[nxt][c0] = b; S.ne[nxt][c0] = b + lf; S.nnode[nxt][c0] = nd.left; S.npar[nxt][c0] = 2 * node;
                    S.nb[nxt][c0 + 1] = b + lf; S.ne[nxt][c0 + 1] = e; S.nnode[nxt][c0 + 1] = nd.right;
                    S.npar[nxt][c0 + 1] = 2 * node + 1;
                    for (int q = 0; q < 3; q++) {
                        S.box[nxt][c0][q] = dkey(INFINITY); S.box[nxt][c0][3 + q] = dkey(-INFINITY);
                        S.box[nxt][c0 + 1][q] = dkey(INFINITY); S.box[nxt][c0 + 1][3 + q] = dkey(-INFINITY);
                    }
                }
                if (node == 0) d.atree[0] = nd;              // the root's box has no parent record to live in
                kd_publish(d.awide, nd, node, par);
            }
            if (tid == 0) S.count[cur] = 0;
            lds_barrier();
            KB_MARK();
            cur = nxt;
            first = false;
        }
        __syncthreads();
        for (int i = tid; i < size; i += KB_T) {
            d.aperm[base + i] = S.id[i];
            s.kx[base + i] = S.x[KB_SW(i)]; s.ky[base + i] = S.y[KB_SW(i)]; s.kz[base + i] = S.z[KB_SW(i)];     // final position order: K1 reads leaves from here
        }
    }
    if (T.seq != 0) kd_publish_tree(T);
}

// ------------------------------------------------------------------------------------------------
// k_kd_top (round 4): the TOP of a tree of up to KT_M members -- every node larger than wave_max -- by ONE workgroup with all
// coordinates in LDS, level by level, instead of two launches per level plus the tail launch.  At N = 4096 (BASELINE config 3) the
// level passes were 37 us of a 78-us build: seven dependent launches of 3-8 us whose workgroups mostly wait for global memory and for
// each other (the chained scan).  Here a level is five LDS phases of one 1024-thread workgroup, four consecutive positions per
// thread, and because the top is cheap it can go further down: the subtrees handed to k_kd_block are half as large (wave_max 768
// instead of 1280 at N = 4096), which halves that kernel's instruction stream per level.
// Same partition as everywhere (see the top of this file), same records; the children's boxes are accumulated while the parent
// is partitioned (as in the level passes), the subtrees' own boxes are k_kd_block's as before.
#ifndef SCA_KT_THREADS
#define SCA_KT_THREADS 1024
#endif
constexpr int KT_M = 4096, KT_T = SCA_KT_THREADS, KT_E = KT_M / KT_T, KT_NODES = 32;
// A node of the top has more than wave_max members, and wave_max >= cap / 2 + 1 >= KD_WAVE_FLOOR + 1 (build_agent_tree_device; SCA_KD_WAVE_CAP
// is clamped to 2 * KD_WAVE_FLOOR): a level of the top therefore holds at most KT_M / (KD_WAVE_FLOOR + 2) nodes, and a thread's KT_E consecutive
// positions lie in at most two of them.  The host checks the same bound with the pass's actual wave_max before it takes this path.
static_assert(KT_M / (KD_WAVE_FLOOR + 2) < KT_NODES, "k_kd_top: a level's nodes must fit KtLds' per-level tables at the smallest wave_max");
static_assert(KT_E <= KD_WAVE_FLOOR + 1, "k_kd_top: a thread's positions must not span more than two nodes");
// Sixteen wavefronts, four consecutive positions per thread.  (Measured with 256 / 512 / 1024 threads, i.e. 16 / 8 / 4 positions each:
// k_kd_top 42 / 30 / 26 us at N = 4096 -- a level is a chain of LDS round trips, and the wavefronts hide each other's.)
#define KT_SW(p) ((((p) & (KT_E - 1)) * KT_T) + ((p) / KT_E))   // position p of thread p / KT_E: the k-th positions of all threads side by side
struct KtLds {
    double x[KT_M], y[KT_M], z[KT_M];       // swizzled: index KT_SW(position)
    int id[KT_M];                           // swizzled
    unsigned short ps[KT_M];                // swizzled: inclusive count of ">= split" members over all positions
    unsigned short mr[KT_M];                // mr[b + m] = position of the member "< split" that is m-th from the right in its node
    int nb[2][KT_NODES], ne[2][KT_NODES], nnode[2][KT_NODES], npar[2][KT_NODES], naxis[KT_NODES];
    double nsplit[KT_NODES];
    unsigned long long box[2][KT_NODES][6];     // the level's node boxes: keys of the minima, keys of the maxima
    unsigned long long cbox[KT_NODES][2][6];    // the two children's, accumulated while the node is partitioned
    int count[2];
    int wtot[KT_T / 64];
};
__global__ __launch_bounds__(KT_T) void k_kd_top(DeviceView d, KdScratch s) {
    SCA_TL(d, TL_KD_TOP);
    SCA_KD_SETPRIO();
    __shared__ KtLds S;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n = d.n;
    for (int i = tid; i < n; i += KT_T) {
        S.x[KT_SW(i)] = s.kx[i]; S.y[KT_SW(i)] = s.ky[i]; S.z[KT_SW(i)] = s.kz[i]; S.id[KT_SW(i)] = d.aperm[i];
    }
    if (tid == 0) {
        S.nb[0][0] = 0; S.ne[0][0] = n; S.nnode[0][0] = 0; S.npar[0][0] = -1; S.count[0] = 1; S.count[1] = 0;
        for (int k = 0; k < 6; k++) S.box[0][0][k] = s.nbox[k];           // the root's, from k_kd_gather
    }
    __syncthreads();
    int cur = 0;
    const int p0 = KT_E * tid;
    for (;;) {
        const int nc = S.count[cur] < KT_NODES ? S.count[cur] : KT_NODES;
        if (nc == 0) break;
        const int nxt = cur ^ 1;
        // ---- split planes of the level's nodes (kdTree.py:85-96); their children's accumulators; the next level's counter
        if (tid == KT_T - 1) S.count[nxt] = 0;                           // (read last at the previous level's top, written next by this level's records)
        if (tid < nc) {
            double mn[3], mx[3];
            for (int q = 0; q < 3; q++) { mn[q] = dunkey(S.box[cur][tid][q]); mx[q] = dunkey(S.box[cur][tid][3 + q]); }
            int axis; double split;
            kd_split(mn, mx, axis, split);
            S.naxis[tid] = axis; S.nsplit[tid] = split;
            for (int sd = 0; sd < 2; sd++)
                for (int q = 0; q < 3; q++) { S.cbox[tid][sd][q] = dkey(INFINITY); S.cbox[tid][sd][3 + q] = dkey(-INFINITY); }
        }
        __syncthreads();
        // ---- the (at most two: a node here has more than wave_max >= 129 members) nodes this thread's KT_E consecutive positions are in
        int sA = -1, bA = 0, eA = 0, sB = -1, bB = 0, eB = 0;
        for (int j = 0; j < nc; j++) {
            const int b = S.nb[cur][j], e = S.ne[cur][j];
            if (b < p0 + KT_E && e > p0) {
                if (sA < 0) { sA = j; bA = b; eA = e; }
                else { sB = j; bB = b; eB = e; }
            }
        }
        const int axA = sA >= 0 ? S.naxis[sA] : 0, axB = sB >= 0 ? S.naxis[sB] : 0;
        const double spA = sA >= 0 ? S.nsplit[sA] : 0.0, spB = sB >= 0 ? S.nsplit[sB] : 0.0;
        const bool full = sA >= 0 && sB < 0 && p0 >= bA && p0 + KT_E <= eA;      // all KT_E in one node: the usual thread
        // ---- flags; the children's boxes
        unsigned gebits = 0, inbits = 0;
        double acc[2][6] = {{INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY}, {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY}};
#pragma unroll
        for (int k = 0; k < KT_E; k++) {
            const int p = p0 + k;
            const bool inA = p >= bA && p < eA, inB = sB >= 0 && p >= bB && p < eB;
            if (inA || inB) {
                const double x = S.x[KT_SW(p)], y = S.y[KT_SW(p)], z = S.z[KT_SW(p)];
                const int axis = inA ? axA : axB;
                const double c = axis == 0 ? x : (axis == 1 ? y : z);
                const bool g = !(c < (inA ? spA : spB));
                inbits |= 1u << k;
                gebits |= g ? (1u << k) : 0u;
                if (full) {
                    // (v_min_f64 / v_max_f64 order -0 below +0 as the keys do)
                    acc[0][0] = __builtin_fmin(acc[0][0], g ? INFINITY : x); acc[0][3] = __builtin_fmax(acc[0][3], g ? -INFINITY : x);
                    acc[0][1] = __builtin_fmin(acc[0][1], g ? INFINITY : y); acc[0][4] = __builtin_fmax(acc[0][4], g ? -INFINITY : y);
                    acc[0][2] = __builtin_fmin(acc[0][2], g ? INFINITY : z); acc[0][5] = __builtin_fmax(acc[0][5], g ? -INFINITY : z);
                    acc[1][0] = __builtin_fmin(acc[1][0], g ? x : INFINITY); acc[1][3] = __builtin_fmax(acc[1][3], g ? x : -INFINITY);
                    acc[1][1] = __builtin_fmin(acc[1][1], g ? y : INFINITY); acc[1][4] = __builtin_fmax(acc[1][4], g ? y : -INFINITY);
                    acc[1][2] = __builtin_fmin(acc[1][2], g ? z : INFINITY); acc[1][5] = __builtin_fmax(acc[1][5], g ? z : -INFINITY);
                } else {
                    const int sl = inA ? sA : sB, sd = g ? 1 : 0;
                    atomicMin(&S.cbox[sl][sd][0], dkey(x)); atomicMin(&S.cbox[sl][sd][1], dkey(y)); atomicMin(&S.cbox[sl][sd][2], dkey(z));
                    atomicMax(&S.cbox[sl][sd][3], dkey(x)); atomicMax(&S.cbox[sl][sd][4], dkey(y)); atomicMax(&S.cbox[sl][sd][5], dkey(z));
                }
            }
        }
        {
            // a 16-lane row's 256 positions sit in one node almost always: twelve row minima of keys, one lane's atomics; else each
            // full lane its own.  All keys as minima (maxima inverted), as in k_kd_block.
            unsigned long long key[2][6];
#pragma unroll
            for (int sd = 0; sd < 2; sd++)
#pragma unroll
                for (int q = 0; q < 3; q++) { key[sd][q] = full ? dkey(acc[sd][q]) : ~0ull; key[sd][3 + q] = full ? ~dkey(acc[sd][3 + q]) : ~0ull; }
            const int rs = row_bcast_i<0>(full ? sA : -1);
            const unsigned long long um = __ballot(full && sA == rs);
            const bool row_uni = ((um >> (lane & 48)) & 0xffffull) == 0xffffull;
            if (row_uni) {
#pragma unroll
                for (int sd = 0; sd < 2; sd++)
#pragma unroll
                    for (int q = 0; q < 6; q++) key[sd][q] = kb_key_min<false>(key[sd][q]);
            }
            if (full && (!row_uni || (lane & 15) == 0)) {
                for (int sd = 0; sd < 2; sd++)
                    for (int q = 0; q < 3; q++) { atomicMin(&S.cbox[sA][sd][q], key[sd][q]); atomicMax(&S.cbox[sA][sd][3 + q], ~key[sd][3 + q]); }
            }
        }
        // ---- block scan of the flags
        const int cnt = __popc(gebits);
        const int incl = wave_incl_scan_i32(cnt);
        if (lane == 63) S.wtot[wid] = incl;
        __syncthreads();
        int excl;
        {
            const int wv = lane < KT_T / 64 ? S.wtot[lane] : 0;
            const int wscan = wave_incl_scan_i32(wv);
            const int wbase = wid > 0 ? __builtin_amdgcn_readlane(wscan, wid - 1) : 0;
            excl = wbase + incl - cnt;
            int run = excl;
#pragma unroll
            for (int k = 0; k < KT_E; k++) { run += (gebits >> k) & 1u; if (p0 + k < n) S.ps[KT_SW(p0 + k)] = (unsigned short)run; }
        }
        __syncthreads();
        // ---- L = #(members < split); the m-th member "< split" of the right part counted from the right
        int pbA = 0, LA = 0, pbB = 0, LB = 0;
        if (sA >= 0) { pbA = bA > 0 ? (int)S.ps[KT_SW(bA - 1)] : 0; LA = (eA - bA) - ((int)S.ps[KT_SW(eA - 1)] - pbA); }
        if (sB >= 0) { pbB = bB > 0 ? (int)S.ps[KT_SW(bB - 1)] : 0; LB = (eB - bB) - ((int)S.ps[KT_SW(eB - 1)] - pbB); }
        {
            int run = excl;
#pragma unroll
            for (int k = 0; k < KT_E; k++) {
                const int p = p0 + k;
                const bool g = (gebits >> k) & 1u;
                run += g ? 1 : 0;
                if ((inbits >> k) & 1u) {
                    const bool inA = p >= bA && p < eA;
                    const int b = inA ? bA : bB, L = inA ? LA : LB, G = run - (inA ? pbA : pbB);
                    if (!g && p >= b + L) S.mr[b + (L - ((p - b + 1) - G))] = (unsigned short)p;
                }
            }
        }
        __syncthreads();
        // ---- the swaps (kdTree.py:108-111) by position; node records and children (kdTree.py:112-122), one lane per node
        {
            int run = excl;
#pragma unroll
            for (int k = 0; k < KT_E; k++) {
                const int p = p0 + k;
                const bool g = (gebits >> k) & 1u;
                run += g ? 1 : 0;
                if (g && ((inbits >> k) & 1u)) {
                    const bool inA = p >= bA && p < eA;
                    const int b = inA ? bA : bB, L = inA ? LA : LB, G = run - (inA ? pbA : pbB);
                    if (p < b + L) {
                        const int q = S.mr[b + G - 1];
                        const int ip = S.id[KT_SW(p)]; S.id[KT_SW(p)] = S.id[KT_SW(q)]; S.id[KT_SW(q)] = ip;
                        double t;
                        t = S.x[KT_SW(p)]; S.x[KT_SW(p)] = S.x[KT_SW(q)]; S.x[KT_SW(q)] = t;
                        t = S.y[KT_SW(p)]; S.y[KT_SW(p)] = S.y[KT_SW(q)]; S.y[KT_SW(q)] = t;
                        t = S.z[KT_SW(p)]; S.z[KT_SW(p)] = S.z[KT_SW(q)]; S.z[KT_SW(q)] = t;
                    }
                }
            }
        }
        if (tid < nc) {
            const int b = S.nb[cur][tid], e = S.ne[cur][tid], node = S.nnode[cur][tid], par = S.npar[cur][tid];
            const int pb = b > 0 ? (int)S.ps[KT_SW(b - 1)] : 0;
            const int L = (e - b) - ((int)S.ps[KT_SW(e - 1)] - pb);
            const int lf = L == 0 ? 1 : L;                               // degenerate: every member on the split plane
            KdNode nd;
            nd.begin = b; nd.end = e; nd.left = node + 1; nd.right = node + 2 * lf;
            for (int q = 0; q < 3; q++) { nd.mn[q] = dunkey(S.box[cur][tid][q]); nd.mx[q] = dunkey(S.box[cur][tid][3 + q]); }
            if (node == 0) d.atree[0] = nd;                              // the root's box has no parent record to live in
            kd_publish(d.awide, nd, node, par);
            const int cbeg[2] = {b, b + lf}, cend[2] = {b + lf, e};
            for (int k = 0; k < 2; k++) {
                KdJob j; j.begin = cbeg[k]; j.end = cend[k]; j.node = k == 0 ? nd.left : nd.right; j.pad = 2 * node + k;
                if (cend[k] - cbeg[k] > s.wave_max) {
                    const int at = atomicAdd(&S.count[nxt], 1);
                    if (at < KT_NODES) {
                        S.nb[nxt][at] = j.begin; S.ne[nxt][at] = j.end; S.nnode[nxt][at] = j.node; S.npar[nxt][at] = j.pad;
                        // (nobody below the midpoint: the box has no extent, both children have the parent's)
                        for (int q = 0; q < 6; q++) S.box[nxt][at][q] = L == 0 ? S.box[cur][tid][q] : S.cbox[tid][k][q];
                    } else atomicOr(&s.counts[KD_MAX_LEVELS + 1], KD_ERR_JOBS);
                } else s.small[atomicAdd(&s.counts[KD_MAX_LEVELS], 1)] = j;
            }
        }
        __syncthreads();
        cur = nxt;
    }
    __syncthreads();
    for (int i = tid; i < n; i += KT_T) {
        d.aperm[i] = S.id[KT_SW(i)];
        s.kx[i] = S.x[KT_SW(i)]; s.ky[i] = S.y[KT_SW(i)]; s.kz[i] = S.z[KT_SW(i)];
    }
}

}  // namespace sca
