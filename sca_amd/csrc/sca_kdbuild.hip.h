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
//   k_kd_gather : coordinates into position order (kx/ky/kz[p] = pos of ids[p]) -> every later pass is coalesced
//   k_kd_level  : one launch per tree level for the nodes with more than KD_WAVE_MAX members, ONE WORKGROUP PER
//                 NODE (block reductions for the box and the counts, ballot-based block scan for the ranks)
//   k_kd_small  : every subtree of <= KD_WAVE_MAX members is finished by ONE WAVEFRONT in LDS, depth first
#pragma once
#include "sca_kernels.hip.h"

namespace sca {

constexpr int KD_WAVE_MAX = 256;       // a node this small is finished (whole subtree) by one wavefront
constexpr int KD_LEVEL_THREADS = 512;
constexpr int KD_SMALL_WAVES = 4;
constexpr int KD_SMALL_STACK = 64;
constexpr int KD_MAX_LEVELS = 40;

struct KdJob { int begin, end, node, pad; };

struct KdScratch {
    double *kx, *ky, *kz;     // [n] coordinates in position order
    int *ml, *mr;             // [n] positions of the k-th misplaced element on the left / right side
    KdJob *jobs[2];           // ping-pong lists of nodes with > KD_WAVE_MAX members
    KdJob *small;             // [n] subtrees handed to k_kd_small
    int *counts;              // [KD_MAX_LEVELS + 2] jobs per level; [KD_MAX_LEVELS] = small count; [KD_MAX_LEVELS+1] = overflow flag
    int job_cap;
};

__global__ __launch_bounds__(256) void k_kd_gather(DeviceView d, KdScratch s) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0) {
        for (int i = 0; i < KD_MAX_LEVELS + 2; i++) s.counts[i] = 0;
        KdJob j; j.begin = 0; j.end = d.n; j.node = 0; j.pad = 0;
        if (d.n > KD_WAVE_MAX) { s.jobs[0][0] = j; s.counts[0] = 1; }
        else { s.small[0] = j; s.counts[KD_MAX_LEVELS] = 1; }
    }
    if (p >= d.n) return;
    const PubRec r = d.rec[d.aperm[p]];
    s.kx[p] = r.px; s.ky[p] = r.py; s.kz[p] = r.pz;
}

__device__ __forceinline__ double wave_min_d(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const double o = __shfl_xor(v, off); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const double o = __shfl_xor(v, off); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// kdTree.py:89-96: split axis and plane from the box
__device__ __forceinline__ void kd_split(const double mn[3], const double mx[3], int &axis, double &split) {
    const double d0 = mx[0] - mn[0], d1 = mx[1] - mn[1], d2 = mx[2] - mn[2];
    axis = (d0 > d1 && d0 > d2) ? 0 : (d1 > d2 ? 1 : 2);
    split = 0.5 * (mx[axis] + mn[axis]);
}

__global__ __launch_bounds__(KD_LEVEL_THREADS) void k_kd_level(DeviceView d, KdScratch s, int level) {
    constexpr int T = KD_LEVEL_THREADS, W = T / 64;
    __shared__ double red[W][6];
    __shared__ int wtot[W];
    __shared__ int itot[W];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const KdJob *in = s.jobs[level & 1];
    KdJob *out = s.jobs[(level + 1) & 1];
    const int njobs = s.counts[level];
    for (int jb = blockIdx.x; jb < njobs; jb += gridDim.x) {
        const KdJob job = in[jb];
        const int b = job.begin, e = job.end;
        // ---- box (kdTree.py:63-83)
        double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int p = b + tid; p < e; p += T) {
            const double x = s.kx[p], y = s.ky[p], z = s.kz[p];
            mn[0] = x < mn[0] ? x : mn[0]; mx[0] = x > mx[0] ? x : mx[0];
            mn[1] = y < mn[1] ? y : mn[1]; mx[1] = y > mx[1] ? y : mx[1];
            mn[2] = z < mn[2] ? z : mn[2]; mx[2] = z > mx[2] ? z : mx[2];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) { mn[k] = wave_min_d(mn[k]); mx[k] = wave_max_d(mx[k]); }
        __syncthreads();                                   // red/wtot free again (previous job)
        if (lane == 0) { for (int k = 0; k < 3; k++) { red[wid][k] = mn[k]; red[wid][3 + k] = mx[k]; } }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double a = red[0][k], c = red[0][3 + k];
            for (int w = 1; w < W; w++) { a = red[w][k] < a ? red[w][k] : a; c = red[w][3 + k] > c ? red[w][3 + k] : c; }
            mn[k] = a; mx[k] = c;
        }
        int axis; double split;
        kd_split(mn, mx, axis, split);
        const double *kc = axis == 0 ? s.kx : (axis == 1 ? s.ky : s.kz);
        // ---- L = #(coord < split)
        int cnt = 0;
        for (int p = b + tid; p < e; p += T) cnt += kc[p] < split ? 1 : 0;
        cnt = wave_sum_i(cnt);
        if (lane == 0) itot[wid] = cnt;
        __syncthreads();
        int L = 0;
        for (int w = 0; w < W; w++) L += itot[w];
        // ---- ranks of the misplaced elements (block scan of the >= split flags, tile by tile)
        int carry = 0, nmis = 0;
        for (int t0 = b; t0 < e; t0 += T) {
            const int p = t0 + tid;
            const bool in_range = p < e;
            const bool ge = in_range && !(kc[p] < split);
            const unsigned long long m = __ballot(ge);
            const int incl_w = __popcll(m & ((2ull << lane) - 1ull));
            __syncthreads();
            if (lane == 0) wtot[wid] = __popcll(m);
            __syncthreads();
            int woff = 0, ttot = 0;
            for (int w = 0; w < W; w++) { const int v = wtot[w]; if (w < wid) woff += v; ttot += v; }
            const int G = carry + woff + incl_w;            // # of >= split in [b, p]
            if (in_range) {
                if (p < b + L) { if (ge) { s.ml[b + G - 1] = p; nmis++; } }
                else if (!ge) { const int lt_incl = (p - b + 1) - G; s.mr[b + (L - lt_incl)] = p; }
            }
            carry += ttot;
        }
        nmis = wave_sum_i(nmis);
        __syncthreads();
        if (lane == 0) itot[wid] = nmis;
        __syncthreads();                                    // also publishes ml / mr inside the workgroup
        int nswap = 0;
        for (int w = 0; w < W; w++) nswap += itot[w];
        // ---- the swaps (kdTree.py:108-111)
        for (int k = tid; k < nswap; k += T) {
            const int p = s.ml[b + k], q = s.mr[b + k];
            const int ip = d.aperm[p], iq = d.aperm[q];
            d.aperm[p] = iq; d.aperm[q] = ip;
            const double xp = s.kx[p], yp = s.ky[p], zp = s.kz[p];
            s.kx[p] = s.kx[q]; s.ky[p] = s.ky[q]; s.kz[p] = s.kz[q];
            s.kx[q] = xp; s.ky[q] = yp; s.kz[q] = zp;
        }
        // ---- node record + children (kdTree.py:112-122)
        if (tid == 0) {
            const int leftSize = L == 0 ? 1 : L;            // degenerate: every member on the split plane
            KdNode nd;
            nd.begin = b; nd.end = e; nd.left = job.node + 1; nd.right = job.node + 2 * leftSize;
            for (int k = 0; k < 3; k++) { nd.mn[k] = mn[k]; nd.mx[k] = mx[k]; }
            d.atree[job.node] = nd;
            KdJob c[2];
            c[0].begin = b; c[0].end = b + leftSize; c[0].node = nd.left; c[0].pad = 0;
            c[1].begin = b + leftSize; c[1].end = e; c[1].node = nd.right; c[1].pad = 0;
            for (int k = 0; k < 2; k++) {
                if (c[k].end - c[k].begin > KD_WAVE_MAX) {
                    if (level + 1 < KD_MAX_LEVELS) {
                        const int at = atomicAdd(&s.counts[level + 1], 1);
                        if (at < s.job_cap) out[at] = c[k]; else s.counts[KD_MAX_LEVELS + 1] = 1;
                    } else s.counts[KD_MAX_LEVELS + 1] = 1;
                } else {
                    const int at = atomicAdd(&s.counts[KD_MAX_LEVELS], 1);
                    s.small[at] = c[k];
                }
            }
        }
    }
}

// One wavefront finishes one subtree of <= KD_WAVE_MAX members, depth first, entirely in LDS.
struct KdSmallLds {
    double x[KD_SMALL_WAVES][KD_WAVE_MAX], y[KD_SMALL_WAVES][KD_WAVE_MAX], z[KD_SMALL_WAVES][KD_WAVE_MAX];
    int id[KD_SMALL_WAVES][KD_WAVE_MAX];
    int ml[KD_SMALL_WAVES][KD_WAVE_MAX], mr[KD_SMALL_WAVES][KD_WAVE_MAX];
    int stk[KD_SMALL_WAVES][KD_SMALL_STACK][3];
};

__global__ __launch_bounds__(KD_SMALL_WAVES * 64) void k_kd_small(DeviceView d, KdScratch s, int levels_run) {
    __shared__ KdSmallLds S;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int njobs = s.counts[KD_MAX_LEVELS];
    // nodes still larger than KD_WAVE_MAX after the last level launch were never split: report, never guess
    if (blockIdx.x == 0 && threadIdx.x == 0 && s.counts[levels_run] > 0) s.counts[KD_MAX_LEVELS + 1] = 1;
    double *X = S.x[wid], *Y = S.y[wid], *Z = S.z[wid];
    int *ID = S.id[wid], *ML = S.ml[wid], *MR = S.mr[wid];
    for (int jb = blockIdx.x * KD_SMALL_WAVES + wid; jb < njobs; jb += gridDim.x * KD_SMALL_WAVES) {
        const KdJob job = s.small[jb];
        const int base = job.begin, size = job.end - job.begin;
        if (size > KD_WAVE_MAX) { if (lane == 0) s.counts[KD_MAX_LEVELS + 1] = 1; continue; }
        for (int i = lane; i < size; i += 64) {
            X[i] = s.kx[base + i]; Y[i] = s.ky[base + i]; Z[i] = s.kz[base + i]; ID[i] = d.aperm[base + i];
        }
        __builtin_amdgcn_wave_barrier();
        int sp = 0;
        int nb = 0, ne = size, nnode = job.node;          // current node, positions relative to base
        bool have = true;
        while (have) {
            have = false;
            // box
            double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int i = nb + lane; i < ne; i += 64) {
                const double x = X[i], y = Y[i], z = Z[i];
                mn[0] = x < mn[0] ? x : mn[0]; mx[0] = x > mx[0] ? x : mx[0];
                mn[1] = y < mn[1] ? y : mn[1]; mx[1] = y > mx[1] ? y : mx[1];
                mn[2] = z < mn[2] ? z : mn[2]; mx[2] = z > mx[2] ? z : mx[2];
            }
#pragma unroll
            for (int k = 0; k < 3; k++) { mn[k] = wave_min_d(mn[k]); mx[k] = wave_max_d(mx[k]); }
            KdNode nd;
            nd.begin = base + nb; nd.end = base + ne; nd.left = 0; nd.right = 0;
            for (int k = 0; k < 3; k++) { nd.mn[k] = mn[k]; nd.mx[k] = mx[k]; }
            int leftSize = 0;
            if (ne - nb > MAX_LEAF) {
                int axis; double split;
                kd_split(mn, mx, axis, split);
                const double *C = axis == 0 ? X : (axis == 1 ? Y : Z);
                int cnt = 0;
                for (int i = nb + lane; i < ne; i += 64) cnt += C[i] < split ? 1 : 0;
                const int L = wave_sum_i(cnt);
                // misplaced on the left, counted from the left
                int carry = 0;
                for (int t0 = nb; t0 < nb + L; t0 += 64) {
                    const int i = t0 + lane;
                    const bool ge = i < nb + L && !(C[i] < split);
                    const unsigned long long m = __ballot(ge);
                    if (ge) ML[carry + __popcll(m & ((1ull << lane) - 1ull))] = i;
                    carry += __popcll(m);
                }
                const int nswap = carry;
                // misplaced on the right, counted from the right
                carry = 0;
                for (int t1 = ne; t1 > nb + L; t1 -= 64) {
                    const int i = t1 - 1 - lane;                 // lane 0 takes the right-most element
                    const bool lt = i >= nb + L && (C[i] < split);
                    const unsigned long long m = __ballot(lt);
                    if (lt) MR[carry + __popcll(m & ((1ull << lane) - 1ull))] = i;
                    carry += __popcll(m);
                }
                __builtin_amdgcn_wave_barrier();
                for (int k = lane; k < nswap; k += 64) {
                    const int p = ML[k], q = MR[k];
                    const int ip = ID[p]; ID[p] = ID[q]; ID[q] = ip;
                    double t;
                    t = X[p]; X[p] = X[q]; X[q] = t;
                    t = Y[p]; Y[p] = Y[q]; Y[q] = t;
                    t = Z[p]; Z[p] = Z[q]; Z[q] = t;
                }
                __builtin_amdgcn_wave_barrier();
                leftSize = L == 0 ? 1 : L;
                nd.left = nnode + 1; nd.right = nnode + 2 * leftSize;
            }
            if (lane == 0) d.atree[nnode] = nd;
            if (leftSize > 0) {
                // right child later, left child now (the order is irrelevant for the result: disjoint ranges)
                if (sp < KD_SMALL_STACK) {
                    if (lane == 0) { S.stk[wid][sp][0] = nb + leftSize; S.stk[wid][sp][1] = ne; S.stk[wid][sp][2] = nd.right; }
                    sp++;
                } else if (lane == 0) s.counts[KD_MAX_LEVELS + 1] = 1;
                ne = nb + leftSize; nnode = nd.left;
                have = true;
            } else if (sp > 0) {
                sp--;
                __builtin_amdgcn_wave_barrier();
                nb = __builtin_amdgcn_readfirstlane(S.stk[wid][sp][0]);
                ne = __builtin_amdgcn_readfirstlane(S.stk[wid][sp][1]);
                nnode = __builtin_amdgcn_readfirstlane(S.stk[wid][sp][2]);
                have = true;
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < size; i += 64) d.aperm[base + i] = ID[i];
    }
}

}  // namespace sca
