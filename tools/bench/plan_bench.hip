// plan_bench.hip -- micro-benchmark of the 3-D Dubins planner's search (sca_dubins.hpp plan3d) on the device: one lane per plan,
// the c4 circle's poses (40-km paths, level flight, headings a few degrees off the goal direction), workgroups of four
// wavefronts as k_track_replan launches them.  Prints ms per launch, candidates per plan and a checksum of the lengths, so that
// variants of the arithmetic (compile-time switches) can be compared for speed AND for identical results.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I sca_amd/csrc tools/bench/plan_bench.hip -o plan_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>
#include <algorithm>
#if defined(SCA_LEAN_STATS) || defined(SCA_LEAN_TIMING)
__device__ unsigned long long g_lean_stats[8];
#endif
#if defined(SCA_LEAN_WAVEITERS)
__shared__ int g_wave_iters[4];
#endif
#include "sca_dubins.hpp"

#ifndef PB_BOUNDS
#define PB_BOUNDS 2
#endif
__global__ __launch_bounds__(256, PB_BOUNDS) void k_plan(const double *q, int n, double *len, int *iters) {
#if defined(SCA_LEAN_WAVEITERS)
    if (threadIdx.x < 4) g_wave_iters[threadIdx.x] = 0;
    const unsigned long long tb = __builtin_readcyclecounter();
#endif
    sca_gm::lds_tables_load();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double qi[5], qf[5];
    for (int k = 0; k < 5; k++) { qi[k] = q[10 * i + k]; qf[k] = q[10 * i + 5 + k]; }
    const double pl[2] = {-M_PI / 4, M_PI / 4};
    const sca_dubins::Plan3D P = sca_dubins::SCA_PLAN3D_LANE(qi, qf, 1.5, pl);
    len[i] = P.length + P.h.r_min * 1e-3 + P.v.t;
    iters[i] = P.iters;
#if defined(SCA_LEAN_WAVEITERS)
    const unsigned long long te = __builtin_readcyclecounter();
    len[i] = (double)(te - tb);                     // (the checksum is void in this build)
    iters[i] = g_wave_iters[threadIdx.x >> 6];
#endif
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 96256;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    std::vector<double> q(10 * (size_t)n);
    const double R = 1.25 * 100000 / (2 * M_PI);
    unsigned s = 12345;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0 / 16777216.0); };
    for (int i = 0; i < n; i++) {
        const double th = 2 * M_PI * i / 100000.0;
        double *p = &q[10 * (size_t)i];
        // a few steps into the episode: 0.3 m along the chord, sideways offsets and heading changes of an avoidance manoeuvre
        p[0] = R * cos(th) + 0.3 * cos(th + M_PI) + (rnd() - 0.5) * 0.2; p[1] = R * sin(th) + 0.3 * sin(th + M_PI) + (rnd() - 0.5) * 0.2; p[2] = 10.0 + (rnd() - 0.5) * 0.1;
        p[3] = fmod(th + M_PI + (rnd() - 0.5) * 0.6, 2 * M_PI); p[4] = (rnd() - 0.5) * 0.3;
        p[5] = -R * cos(th); p[6] = -R * sin(th); p[7] = 10.0; p[8] = fmod(th + M_PI, 2 * M_PI); p[9] = 0.0;
    }
    double *dq, *dl; int *di;
    hipMalloc(&dq, q.size() * 8); hipMalloc(&dl, n * 8); hipMalloc(&di, n * 4);
    hipMemcpy(dq, q.data(), q.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < reps + 1; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_plan, dim3((n + 255) / 256), dim3(256), 0, 0, dq, n, dl, di);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best) best = ms;
    }
    std::vector<double> l(n); std::vector<int> it(n);
    if (argc > 4) {
        // arrangement experiments: the plans re-ordered by their (now known) candidate counts; argv[4] = asc | desc | lpt | rand
        hipMemcpy(it.data(), di, n * 4, hipMemcpyDeviceToHost);
        std::vector<int> order(n);
        for (int i = 0; i < n; i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return it[a] < it[b]; });      // ascending
        std::vector<int> perm;
        const std::string mode = argv[4];
        const int W = (n + 255) / 256, D = W > 256 ? W - 256 : 0;                  // workgroups, and how many of them double up a CU
        if (mode == "asc") perm = order;
        else if (mode == "desc") perm.assign(order.rbegin(), order.rend());
        else if (mode == "rand") { perm = order; unsigned s2 = 99; for (int i = n - 1; i > 0; i--) { s2 = s2 * 1664525u + 1013904223u; std::swap(perm[i], perm[(s2 >> 4) % (i + 1)]); } }
        else {                                                                      // lpt: [short half | long | short half]
            const int nshort = std::min(n, 2 * D * 256), half = (nshort / 2 / 256) * 256;
            for (int i = 0; i < half; i++) perm.push_back(order[i]);
            for (int i = nshort; i < n; i++) perm.push_back(order[i]);
            for (int i = half; i < nshort; i++) perm.push_back(order[i]);
        }
        std::vector<double> q2(q.size());
        for (int i = 0; i < n; i++) memcpy(&q2[10 * (size_t)i], &q[10 * (size_t)perm[i]], 80);
        hipMemcpy(dq, q2.data(), q2.size() * 8, hipMemcpyHostToDevice);
        best = 1e9f;
        for (int r = 0; r < reps + 1; r++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_plan, dim3((n + 255) / 256), dim3(256), 0, 0, dq, n, dl, di);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r > 0 && ms < best) best = ms;
        }
    }
    hipMemcpy(l.data(), dl, n * 8, hipMemcpyDeviceToHost); hipMemcpy(it.data(), di, n * 4, hipMemcpyDeviceToHost);
#if defined(SCA_LEAN_WAVEITERS)
    { double cy = 0, itw = 0, mxc = 0, mxi = 0; int nw = 0;
      for (int i = 0; i < n; i += 64) { double c = 0; for (int k = i; k < i + 64 && k < n; k++) c = std::max(c, l[k]); cy += c; itw += it[i]; mxc = std::max(mxc, c); mxi = std::max(mxi, (double)it[i]); nw++; }
      printf("  waves %d: mean %.0f cycles, %.1f wave-iterations (%.0f cycles each); max %.0f cycles, max %.0f iterations\n", nw, cy / nw, itw / nw, cy / itw, mxc, mxi); }
#endif
    double sum = 0; long its = 0; int mx = 0; unsigned long long h = 1469598103934665603ull;
    for (int i = 0; i < n; i++) { sum += l[i]; its += it[i]; if (it[i] > mx) mx = it[i]; unsigned long long b; memcpy(&b, &l[i], 8); h = (h ^ b) * 1099511628211ull; }
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void *)k_plan);
    printf("%-28s n %d  %.4f ms  candidates/plan %.1f (max %d)  hash %016llx  vgprs %d scratch %d\n", argc > 3 ? argv[3] : "", n, best, (double)its / n, mx, h,
           fa.numRegs, (int)fa.localSizeBytes);
#if defined(SCA_LEAN_TIMING)
    { unsigned long long st[8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_lean_stats), sizeof st);
    const double it = (double)st[5];
    printf("  wave-candidates %.0f; cycles per candidate: vc %.0f  H words %.0f  V frame-a %.0f  V sincos %.0f  V words %.0f\n", it, st[0] / it, st[1] / it, st[2] / it, st[3] / it, st[4] / it);
    printf("  search loop per lane: max %llu cycles, lane-0 sum %llu cycles (launches included: all)\n", st[6], st[7]); }
#endif
#if defined(SCA_LEAN_STATS)
    unsigned long long st[8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_lean_stats), sizeof st);
    printf("  lean stats: fast %llu  fast-then-general %llu  general %llu | odd %llu !dom %llu !far_theta %llu !far_dV %llu\n", st[0], st[1], st[2], st[3], st[4], st[5], st[6]);
#endif
    return 0;
}
