// sca_grid.hip.h -- SCA_NBR_GRID: neighbour selection through a uniform hashed grid instead of the reference's kd-tree.
//
// Why it exists: the kd-tree of kdTree.py:56-122 is a chain of ~17 dependent tree levels over ALL agents,
// so on several GPUs every rank repeats it in full.  A counting sort of the agents into cells of neighborDist is three short
// launches whatever the swarm looks like, and a query reads 27 cells.
//
// What it returns.  Agent.insertAgentNeighbor (agent.py:79-99) keeps every object with distSq < rangeSq in a list sorted by
// distSq; which objects those are does not depend on the visit order as long as no more than maxNeighbors are in range, so
// for such agents the grid's list holds the same (object, distSq) pairs as the reference's.  Two things do depend on the
// kd-tree's visit order and are therefore NOT reproduced:
//   * the order of entries with EQUAL rounded distSq (here: obstacles first, in obstacle-tree order, then agents by id);
//     it matters to nobody but the LP of orca3dPolicyOfficial.py, whose plane order is the list order;
//   * which 16 survive when more than 16 are in range (agent.py:87-90 evicts by visit order): the grid keeps
//     the 16 nearest and raises SCA_ST_NBR_OVERFLOW in the agent's status word.
// The collision rule (first colliding object clears the list, afterwards only colliding objects are admitted, agent.py:82-99)
// leaves "all colliding objects in range" whatever the order: reproduced exactly.
// Obstacles keep their kd-tree (built once on the host, kdTree.py:158-227): the obstacle part of a list is the reference's.
//
//   k_grid_count : one lane per agent: cell key -> bucket, arrival slot by atomicAdd (+ the per-agent prologue of the solver)
//   k_grid_alloc : one lane per bucket: a range of sorted positions per non-empty bucket (block scan + one atomic per block;
//                  the ranges need no particular order)
//   k_grid_fill  : one lane per agent: coordinates, id and cell key into bucket order
//   k_neighbors_grid : four agents per wavefront, 16 lanes each (one DPP row): 27 probes, members 16 at a time, the bounded
//                  sorted list one entry per lane
#pragma once
#include "sca_kernels.hip.h"
#include "sca_kdbuild.hip.h"      // kd_answer_listed / auto_arrive_second (the launch-free form of SCA_NBR_AUTO's kd query)

namespace sca {

struct GridDev {
    int *count;                   // [H] members per bucket; all zero between builds (k_grid_alloc clears what it reads)
    int2 *range;                  // [H] first sorted position and member count of the bucket
    int *cursor;                  // [1] sorted positions handed out so far
    int *bucket;                  // [n] bucket of agent i
    int *slot;                    // [n] arrival slot of agent i inside its bucket
    double *gx, *gy, *gz;         // [n] coordinates in bucket order
    int *gid;                     // [n] agent ids in bucket order
    unsigned long long *gkey;     // [n] cell keys in bucket order: several cells can share a bucket
    int hbits;                    // H = 1 << hbits
    int skip_prep;                // 1: the prologue of the tracker's agents is left to the tracker's kernels (v_pref still being computed); 2: nobody's
    double inv_cell;              // 1 / cell size; the cell is a little larger than neighborDist (see grid_inv_cell)
};

// Every object with distSq < rangeSq must sit in one of the 27 cells around the agent's.  Membership is decided on distSq ROUNDED
// to 5 decimals (util.py:100): round(d^2, 5) < nd^2 admits d^2 < nd^2 + 0.5e-5 when nd^2 is not on the 1e-5 grid, i.e.
// d < sqrt(nd^2 + 1e-5) with room to spare -- for a small neighbor_dist that is relatively far beyond nd (nd = 0.1: 5e-4),
// so the cell is derived from that bound, not from a relative slack; the factor covers the two roundings of floor(x * inv_cell).
__host__ __device__ inline double grid_inv_cell(double neighbor_dist) { return 1.0 / (sqrt(neighbor_dist * neighbor_dist + 1.0e-5) * (1.0 + 1.0e-9)); }

__device__ __forceinline__ long long grid_cell(double x, double inv_cell) { return (long long)floor(x * inv_cell); }
__device__ __forceinline__ unsigned long long grid_key(long long cx, long long cy, long long cz) {
    // 21 bits per axis: two cells alias only if they are 2^21 cells apart, never two of the 27 around one agent
    return ((unsigned long long)(cx & 0x1fffff) << 42) | ((unsigned long long)(cy & 0x1fffff) << 21) | (unsigned long long)(cz & 0x1fffff);
}
__device__ __forceinline__ int grid_bucket(unsigned long long key, int hbits) {
    return (int)((key * 0x9E3779B97F4A7C15ull) >> (64 - hbits));
}

__global__ __launch_bounds__(256) void k_grid_count(DeviceView d, GridDev g, Params P) {
    SCA_TL(d, TL_GRID_COUNT);
    SCA_KD_SETPRIO();                                                    // (see sca_kdbuild.hip.h: short launches beside the re-plan kernel)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 256) d.done_count[i * 32] = 0;                                   // start of a step: K4's counters
    if (i == 0) { *d.fb_count = 0; *g.cursor = 0; if (d.kdq_count) *d.kdq_count = 0; }   // ... an empty fallback list, no position handed out, nobody for the kd query
    if (i >= present_count(d)) return;
    const int a = present_agent(d, i);                                        // (bucket / slot are indexed by the position in the list)
    const PubRec r = d.rec[a];
    const unsigned long long key = grid_key(grid_cell(r.px, g.inv_cell), grid_cell(r.py, g.inv_cell), grid_cell(r.pz, g.inv_cell));
    const int h = grid_bucket(key, g.hbits);
    g.bucket[i] = h;
    g.slot[i] = atomicAdd(&g.count[h], 1);
    const bool mine = d.present ? i < shard_size(d) : shard_owns(d, a);
    if (mine && g.skip_prep != 2 && !(g.skip_prep && tracker_owns(d, a))) prep_agent<PREP_LIBM>(d, P, (Prep *)d.prep, a);
}

// GRID_ALLOC_PER buckets per lane: one atomic on the cursor per 2048 buckets (an atomic per 256 buckets, 1024 of them on one
// address at N = 100 000, cost 13 us of the kernel's 14: the same-address rate is ~12 ns per atomic)
constexpr int GRID_ALLOC_PER = 8;
__global__ __launch_bounds__(256) void k_grid_alloc(GridDev g) {
    SCA_KD_SETPRIO();                                                    // (see sca_kdbuild.hip.h: short launches beside the re-plan kernel)
    __shared__ int wtot[4];
    __shared__ int base_sh;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int h0 = (blockIdx.x * 256 + tid) * GRID_ALLOC_PER;
    const int H = 1 << g.hbits;
    int c[GRID_ALLOC_PER], mine = 0;
#pragma unroll
    for (int q = 0; q < GRID_ALLOC_PER; q++) { c[q] = h0 + q < H ? g.count[h0 + q] : 0; mine += c[q]; }
    // inclusive scan over the wavefront (row shifts, then the row totals), then over the four wavefronts
    int v = mine, t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); v += t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); v += t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); v += t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); v += t;
    const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47);
    const int row = lane >> 4;
    const int incl = v + (row > 0 ? r0 : 0) + (row > 1 ? r1 : 0) + (row > 2 ? r2 : 0);
    if (lane == 63) wtot[wid] = incl;
    __syncthreads();
    int woff = 0, total = 0;
    for (int w = 0; w < 4; w++) { const int x = wtot[w]; if (w < wid) woff += x; total += x; }
    if (tid == 0) base_sh = total ? atomicAdd(g.cursor, total) : 0;
    __syncthreads();
    int at = base_sh + woff + incl - mine;
#pragma unroll
    for (int q = 0; q < GRID_ALLOC_PER; q++) {
        if (h0 + q < H) {
            g.range[h0 + q] = make_int2(at, c[q]);
            if (c[q]) g.count[h0 + q] = 0;                                    // ready for the next build
        }
        at += c[q];
    }
}

__global__ __launch_bounds__(256) void k_grid_fill(DeviceView d, GridDev g) {
    SCA_TL(d, TL_GRID_FILL);
    SCA_KD_SETPRIO();                                                    // (see sca_kdbuild.hip.h: short launches beside the re-plan kernel)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= present_count(d)) return;
    const int a = present_agent(d, i);
    const PubRec r = d.rec[a];
    const int at = g.range[g.bucket[i]].x + g.slot[i];
    g.gx[at] = r.px; g.gy[at] = r.py; g.gz[at] = r.pz;
    g.gid[at] = a;
    g.gkey[at] = grid_key(grid_cell(r.px, g.inv_cell), grid_cell(r.py, g.inv_cell), grid_cell(r.pz, g.inv_cell));
}

__device__ __forceinline__ double shfl_d(double x, int src) {
    const unsigned long long v = (unsigned long long)__double_as_longlong(x);
    const int lo = __shfl((int)(unsigned)v, src), hi = __shfl((int)(unsigned)(v >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) {
    const int lo = __shfl((int)(unsigned)v, src), hi = __shfl((int)(unsigned)(v >> 32), src);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

// probe q (0..26) of the 27 cells around (cx, cy, cz)
__device__ __forceinline__ unsigned long long grid_probe_key(long long cx, long long cy, long long cz, int q) {
    const int dx = q % 3 - 1, dy = (q / 3) % 3 - 1, dz = q / 9 - 1;
    return grid_key(cx + dx, cy + dy, cz + dz);
}

// K1 on the grid: FOUR AGENTS PER WAVEFRONT, 16 lanes each (the layout of k_neighbors_kd4).  Obstacles first, through
// their kd-tree exactly as in the kd kernels (scaPolicy.py:114-116, agent.py:101-124); then the agents of the 27 cells.
// Every agent that is not done gets its collision candidates for K4 (`near`), also on the bootstrap step, when the
// reference builds no list (scaPolicy.py:34): K4 then never needs a tree.
// AUTO (SCA_NBR_AUTO): a list the grid cannot give exactly -- it overflowed, or two neighbours of one kind have the same rounded distance
// (the reference orders those by the kd-tree's visit order, agent.py:87-90: append, stable sort) -- is not flagged but handed to the kd
// query (d.kdq_list); every other list IS the reference's, entry for entry: a sorted list without ties has one order.
// (the kernel's body; `listed` comes back true for the lanes of a group whose agent went onto the kd query's list)
template <bool AUTO, bool HAS_OBS>
__device__ __forceinline__ void neighbors_grid_body(const DeviceView &d, const GridDev &g, const Params &P, double agent_reach,
                                                    double obs_reach, double max_radius, bool &listed) {
    __shared__ int stacks[K1P_WAVES][K1P_APW][KD_STACK];
    __shared__ int pfx[K1P_WAVES][K1P_APW][32], fpos[K1P_WAVES][K1P_APW][32];
    __shared__ unsigned long long pkey[K1P_WAVES][K1P_APW][32];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int grp = lane >> 4, gl = lane & 15, gshift = grp << 4;
    const int idx = (blockIdx.x * K1P_WAVES + wid) * K1P_APW + grp;
    const int scnt = shard_size(d);
    if (scnt <= 0) return;
    const bool exists = idx < scnt;
    const int agent = shard_agent(d, exists ? idx : scnt - 1);           // clamp: idle groups read a valid record, write nothing
    const PubRec me = d.rec[agent];
    int st = 0;
    const bool done = (me.flags & (FLAG_AT_GOAL | FLAG_COLLISION | FLAG_TIMEOUT)) != 0;   // mampenv.py:35
    const int pol = d.policy[agent];
    const bool orca = (pol == POL_ORCA || pol == POL_ORCA_LP);
    F3 vA; vA.x = me.vx; vA.y = me.vy; vA.z = me.vz;
    const bool bootstrap = !orca && l3norm_f32zero(vA, false) <= 1e-5;   // scaPolicy.py:34: no computeNeighbors on the bootstrap step
    const bool scan = exists && !done;
    const bool want_list = scan && !bootstrap;
    const V3 pA = v3(me.px, me.py, me.pz);
    const double rangeSq = d.ap ? d.ap[agent].range_sq : P.range_sq;         // scaPolicy.py:112: neighborDist ** 2, the agent's own where the swarm is heterogeneous (a cell is the LARGEST)
    const int maxn = d.ap ? d.ap[agent].max_neighbors : P.max_neighbors;
    const double reach_a = me.radius + agent_reach, reach_o = me.radius + obs_reach;
    const double rmax2 = (me.radius + max_radius) * (me.radius + max_radius);
    const int tk = gl >= 2 ? (gl - 2) >> 2 : 0;                     // box term of this lane (see k_neighbors_kd4)
    const bool t_is_mx = gl >= 2 && (((gl - 2) >> 1) & 1);
    const bool t_live = gl >= 2 && gl < 14;
    const double pk = tk == 0 ? pA.x : (tk == 1 ? pA.y : pA.z);
    double Ld = 0.0; int Li = -1; int cnt = 0; bool coll = false; int near_cnt = 0;
    int *near_out = d.near_id + (size_t)agent * NEAR_MAX;
    int *stack = stacks[wid][grp];

    // one object in range of the group's agent: the list stays sorted by (distSq, obstacles before agents, id); a full list
    // keeps its 16 smallest and says so
    auto member = [&](bool act, bool cb, double db, int ib) {
        if (act && cb && !coll) { coll = true; cnt = 0; }                            // agent.py:83-85
        bool ins = act && (cb || !coll);
        const bool e_lt = gl < cnt && (Ld < db || (Ld == db && ((Li & NBR_OBSTACLE_BIT) != 0 || ((ib & NBR_OBSTACLE_BIT) == 0 && Li < ib))));
        const unsigned bm = (unsigned)((__ballot(ins && e_lt) >> gshift) & 0xffffull);
        const int pos = __popc(bm);
        const bool full = ins && cnt == maxn;
        if (full) { st |= ST_NBR_OVERFLOW; if (pos >= maxn) ins = false; }
        const int ncnt = (ins && full) ? cnt - 1 : cnt;
        const double up_d = row_shr1_d(Ld);
        const int up_i = row_shr1_i(Li);
        if (ins) {
            if (gl > pos && gl <= ncnt) { Ld = up_d; Li = up_i; }
            if (gl == pos) { Ld = db; Li = ib; }
            cnt = ncnt + 1;
        }
    };
    // the <= 16 objects a group has just measured, lane by lane
    auto insert_all = [&](unsigned todo, bool c, double dsq, int o) {
        const int ci = c ? 1 : 0;
        while (__any(todo != 0)) {
            const bool act = todo != 0;
            const int b = act ? __ffs((int)todo) - 1 : 0;
            todo &= todo - 1;
            const int src = gshift + b;
            const bool cb = __shfl(ci, src) != 0;
            const double db = shfl_d(dsq, src);
            const int ib = __shfl(o, src);
            member(act, cb, db, ib);
        }
    };

    // ---- obstacles: kd-tree of kdTree.py:232-262, as in k_neighbors_kd4
    if (HAS_OBS && d.m > 0) {
        const double *wd = (const double *)d.owide;
        int node = 0, sp = 0;
        bool have = scan;
        while (__any(have)) {
            const double w = have ? wd[(size_t)node * 16 + gl] : 0.0;
            const double h0 = row_bcast_d<0>(w), h1 = row_bcast_d<1>(w);
            const int nb = lo32(h0), ne = hi32(h0), nl = lo32(h1), nr_ = hi32(h1);
            const bool leaf = have && (ne - nb <= MAX_LEAF);
            const bool inner = have && !leaf;
            bool descend = false; int next = 0;
            if (__any(inner)) {
                double t = t_is_mx ? pk - w : w - pk;
                t = fmax(0.0, t);
                const double sq = t_live ? t * t : 0.0;
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
                if (valid) {                                         // agent.py:101-124
                    o = d.operm[nb + gl];
                    const ObsRec orec = d.obs_sorted[nb + gl];
                    const V3 pO = v3(orec.px, orec.py, orec.pz);
                    const double distSq1 = l3normsq(pA, pO);
                    const double tt = l3norm(pA, pO) - orec.radius;
                    dsq = m_pow2(tt);                        // (... ) ** 2 = libm's pow (agent.py:106)
                    const double rs = me.radius + orec.radius;
                    r = dsq < rangeSq;
                    c = r && distSq1 < rs * rs;
                    const V3 dd = pA - pO;
                    nr = (dd.x * dd.x + dd.y * dd.y + dd.z * dd.z) < reach_o * reach_o;
                    o |= NBR_OBSTACLE_BIT;
                }
                {
                    const unsigned nm = (unsigned)((__ballot(nr) >> gshift) & 0xffffull);
                    if (nr) { const int at = near_cnt + __popc(nm & ((1u << gl) - 1u)); if (at < NEAR_MAX) near_out[at] = o; }
                    near_cnt += __popc(nm);
                }
                insert_all((unsigned)((__ballot(r && want_list) >> gshift) & 0xffffull), c, dsq, o);
            }
            if (have) {
                if (descend) node = next;
                else if (sp > 0) { sp--; node = stack[sp]; }
                else have = false;
            }
        }
    }

    // ---- agents: the 27 cells around the agent's (agent.py:79-99).  Lane gl probes cells gl and gl + 16; the members of ALL
    // non-empty probes are then taken 16 at a time as ONE flat sequence (round 4: probe by probe -- one dependent trip to memory per
    // non-empty cell, ~14 of them for a sparse 3-D swarm with ~21 members in its 27 cells -- took 27 us at N = 4096; flat: two
    // rounds).  The flat index -> (probe, member) map is a 32-entry prefix table per group in LDS, searched by bisection.
    {
        const long long cx = grid_cell(pA.x, g.inv_cell), cy = grid_cell(pA.y, g.inv_cell), cz = grid_cell(pA.z, g.inv_cell);
        const unsigned long long key_a = grid_probe_key(cx, cy, cz, gl);
        const unsigned long long key_b = grid_probe_key(cx, cy, cz, gl + 16 < 27 ? gl + 16 : 26);
        int2 ra = make_int2(0, 0), rb = make_int2(0, 0);
        if (scan) {
            ra = g.range[grid_bucket(key_a, g.hbits)];
            if (gl + 16 < 27) rb = g.range[grid_bucket(key_b, g.hbits)];
        }
        // exclusive prefix of the member counts in probe order (probes 0..15 on the lanes' first slot, 16..26 on their second)
        auto row_incl = [](int v) {
            int t;
            t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); v += t;
            t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); v += t;
            t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); v += t;
            t = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); v += t;
            return v;
        };
        const int ia = row_incl(ra.y), ib = row_incl(rb.y);
        const int tot_a = row_bcast_i<15>(ia), tot_b = row_bcast_i<15>(ib);
        const int total = tot_a + tot_b;
        int *pf = pfx[wid][grp];
        int *fp = fpos[wid][grp];
        unsigned long long *kq = pkey[wid][grp];
        pf[gl] = ia - ra.y; fp[gl] = ra.x; kq[gl] = key_a;
        pf[16 + gl] = gl + 16 < 27 ? tot_a + ib - rb.y : 0x7fffffff;      // (entries 27..31: sentinels above every index)
        fp[16 + gl] = rb.x; kq[16 + gl] = key_b;
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        for (int base = 0; __any(base < total); base += 16) {
            const int f = base + gl;
            const bool valid = f < total;
            int o = -1; double dsq = 0.0; bool c = false, r = false, nr = false;
            if (valid) {
                int q = 0;                                               // the largest q with pf[q] <= f (empty probes have zero width)
#pragma unroll
                for (int stp = 16; stp >= 1; stp >>= 1) if (pf[q + stp] <= f) q += stp;
                const int at = fp[q] + (f - pf[q]);
                o = g.gid[at];
                if (g.gkey[at] == kq[q] && o != agent) {                 // another cell of the same bucket: not this probe's
                    dsq = l3normsq(pA, v3(g.gx[at], g.gy[at], g.gz[at]));
                    r = dsq < rangeSq;
                    if (r && dsq < rmax2) { const double rs = me.radius + d.rec[o].radius; c = dsq < rs * rs; }
                    nr = dsq < reach_a * reach_a;
                }
            }
            {
                const unsigned nm = (unsigned)((__ballot(nr) >> gshift) & 0xffffull);
                if (nr) { const int at = near_cnt + __popc(nm & ((1u << gl) - 1u)); if (at < NEAR_MAX) near_out[at] = o; }
                near_cnt += __popc(nm);
            }
            insert_all((unsigned)((__ballot(r && want_list) >> gshift) & 0xffffull), c, dsq, o);
        }
    }
    if (!exists) return;
    const bool complete = near_cnt <= NEAR_MAX && reach_a * reach_a <= rangeSq && reach_o * reach_o <= rangeSq;
    if (!want_list) {
        d.nbr_id[agent * K_MAX + gl] = -1; d.nbr_dsq[agent * K_MAX + gl] = 0.0;
        if (gl == 0) {
            d.coll_new[agent] = 0; d.nbr_valid[agent] = 0; d.nbr_n[agent] = 0; d.status[agent] = 0;
            d.near_n[agent] = (scan && complete) ? near_cnt : -1;
        }
        return;
    }
    if (AUTO) {
        const double nd = row_shl_d<1>(Ld);                              // (every lane executes the cross-lane reads)
        const int ni = __builtin_amdgcn_mov_dpp(Li, 0x101, 0xf, 0xf, true);
        const bool tie = gl + 1 < cnt && nd == Ld && ((ni ^ Li) & NBR_OBSTACLE_BIT) == 0;
        const bool again = (((__ballot(tie) >> gshift) & 0xffffull) != 0) || (st & ST_NBR_OVERFLOW) != 0;
        if (again) st &= ~ST_NBR_OVERFLOW;                               // (the kd query writes the agent's status afresh)
        // one atomic per wavefront for its (up to four) listed agents -- and none once the count has passed the cap: "too many" is
        // all the later passes and the gated kd query of everybody need to know (100 000 same-address atomics were 260 us)
        const unsigned long long am = __ballot(again && gl == 0);
        if (am != 0) {
            const int leader = __ffsll((long long)am) - 1;
            int base = 0;
            if ((int)(threadIdx.x & 63) == leader) {
                const int seen = __atomic_load_n(d.kdq_count, __ATOMIC_RELAXED);        // (a stale value only costs an atomic)
                base = seen > d.kdq_cap ? -1 : atomicAdd(d.kdq_count, __popcll(am));
                if ((__atomic_load_n(d.kdq_busy, __ATOMIC_RELAXED) & 1u) == 0) atomicOr(d.kdq_busy, 1u);   // somebody is listed: the pass waits for the kd query
            }
            base = __shfl(base, leader);
            if (again && gl == 0 && base >= 0) {
                const int at = base + __popcll(am & ((1ull << (threadIdx.x & 63)) - 1ull));
                if (at < d.n) d.kdq_list[at] = agent;
            }
            listed = true;                                               // (the whole wavefront: its workgroup fences before it reports)
        }
    }
    d.nbr_id[agent * K_MAX + gl] = (gl < cnt) ? Li : -1;
    d.nbr_dsq[agent * K_MAX + gl] = (gl < cnt) ? Ld : 0.0;
    if (gl == 0) {
        d.nbr_n[agent] = cnt;
        d.nbr_valid[agent] = 1;
        d.coll_new[agent] = coll ? 1u : 0u;
        d.status[agent] = st;
        d.near_n[agent] = complete ? near_cnt : -1;
    }
}

template <bool AUTO, bool HAS_OBS>
__global__ __launch_bounds__(K1P_WAVES * 64) void k_neighbors_grid(DeviceView d, GridDev g, Params P, double agent_reach,
                                                                  double obs_reach, double max_radius) {
    SCA_TL(d, TL_NBR_GRID);
    SCA_K1_SETPRIO();
    bool listed = false;
    neighbors_grid_body<AUTO, HAS_OBS>(d, g, P, agent_reach, obs_reach, max_radius, listed);
    if (AUTO && d.auto_sync) {
        // The launch-free form of the kd query (KdTail, sca_kdbuild.hip.h): this launch's last workgroup -- by ticket -- knows the list's
        // final length.  Zero: nothing to do, nobody fenced, nobody waits.  Otherwise it waits for the pass's tree (whose build is ahead
        // of this launch in the host's order) and answers the listed agents here, before the launch ends: the solve behind it needs no
        // stream wait.  A workgroup that listed somebody first writes back what it stored -- the list entries, and the lists the kd query
        // is about to overwrite, possibly from another XCD -- before its ticket says "through".
        __shared__ int grid_listed;
        __shared__ double rst[K1P_WAVES * KD_RSTACK][16];
        if (__syncthreads_or(listed ? 1 : 0)) __threadfence();
        else __syncthreads();
        if (threadIdx.x == 0) {
            int n = 0;
            if (__hip_atomic_fetch_add(d.auto_sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
                __hip_atomic_store(d.auto_sync + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                n = __hip_atomic_load(d.kdq_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (d.kdq_stats) {                                       // statistics, as k_neighbors_kd_auto keeps them
                    d.kdq_stats[0] += 1; d.kdq_stats[1] += (unsigned long long)n;
                    if ((unsigned long long)n > d.kdq_stats[2]) d.kdq_stats[2] = (unsigned long long)n;
                    if (n > 0) d.kdq_stats[3] += 1;
                }
                if (n > 0) {
                    int spins = 0;                                       // the pass's tree: usually long complete
                    while ((int)(__hip_atomic_load(d.auto_sync + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - d.auto_pass_seq) < 0) {
                        __builtin_amdgcn_s_sleep(16);
                        if (++spins > (1 << 24)) { if (d.auto_err) atomicOr(d.auto_err, KD_ERR_SPIN); break; }   // (tens of seconds: the build died)
                    }
                    __threadfence();                                     // the tree, the other workgroups' list entries: read afresh
                }
            }
            grid_listed = n;
        }
        __syncthreads();
        if (grid_listed > 0) {
            kd_answer_listed<K1P_WAVES, HAS_OBS>(d, P, agent_reach, obs_reach, max_radius, grid_listed, rst);
            __syncthreads();
            if (threadIdx.x == 0) atomicAnd(d.kdq_busy, ~1u);            // (nobody waits for the bit in this form; kept consistent)
        }
    }
}

// K4's fallback for an agent without a candidate list (more than NEAR_MAX objects within reach): the whole wavefront looks
// through the 27 cells around the agent's OLD position (the grid holds the step's old positions).
__device__ __forceinline__ bool collide_scan_grid(const DeviceView &d, const GridDev &g, const CollideCtx &c, int agent, int lane) {
    const long long cx = grid_cell(c.p_old.x, g.inv_cell), cy = grid_cell(c.p_old.y, g.inv_cell), cz = grid_cell(c.p_old.z, g.inv_cell);
    const unsigned long long key = grid_probe_key(cx, cy, cz, lane < 27 ? lane : 26);
    int2 r = make_int2(0, 0);
    if (lane < 27) r = g.range[grid_bucket(key, g.hbits)];
    bool hit = false;
    unsigned long long pm = __ballot(r.y > 0);
    while (pm) {
        const int q = __ffsll((long long)pm) - 1;
        pm &= pm - 1;
        const int first_pos = __builtin_amdgcn_readlane(r.x, q), members = __builtin_amdgcn_readlane(r.y, q);
        const unsigned long long kq = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(key >> 32), q) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)(unsigned)key, q);
        for (int off = lane; off < members; off += 64) {
            const int at = first_pos + off;
            const int j = g.gid[at];
            if (g.gkey[at] == kq && j != agent) hit = hit || collide_agent(d, c, j);
        }
    }
    return __ballot(hit) != 0;
}

__global__ __launch_bounds__(K4_WAVES * 64) void k_collide_finish_grid(DeviceView d, GridDev g, Params P, double agent_reach,
                                                                     double obs_reach, int check_arrived) {
    SCA_TL(d, TL_COLLIDE);
    __shared__ int stacks[K4_WAVES][KD_STACK];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    collide_finish_body(d, P, check_arrived, [&](int ag, bool obs_only) {
        bool r = collide_traverse(d, agent_reach, obs_reach, stacks[wid], ag, lane, true);      // obstacles: their kd-tree
        if (!obs_only && !r) {
            PubRec me_old;
            const CollideCtx c = collide_ctx(d, ag, me_old);
            r = collide_scan_grid(d, g, c, ag, lane);
        }
        return r;
    });
}

}  // namespace sca
