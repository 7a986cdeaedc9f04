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
#include <map>
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
    const unsigned long long wb = wall_clock64();
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
    const int ln = threadIdx.x & 63;                // (the checksum is void in this build)
    len[i] = ln == 0 ? (double)(te - tb) : ln == 3 ? (double)(wall_clock64() - wb) : ln == 4 ? (double)wb : (ln == 1 ? (double)tb : (double)(__builtin_amdgcn_s_getreg((31 << 11) | 4) | (__builtin_amdgcn_s_getreg((31 << 11) | 20) << 24)));
    iters[i] = g_wave_iters[threadIdx.x >> 6];
#endif
}

int main(int argc, char **argv) {
    // argv[1]: the number of synthetic plans, or a file of n x 10 doubles (qi[5], qf[5]: recorded poses, scratch/dump_c4.py)
    int n = argc > 1 ? atoi(argv[1]) : 96256;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    std::vector<double> q;
    FILE *pf = (argc > 1 && n == 0) ? fopen(argv[1], "rb") : nullptr;
    if (pf) {
        fseek(pf, 0, SEEK_END); n = (int)(ftell(pf) / 80); fseek(pf, 0, SEEK_SET);
        q.resize(10 * (size_t)n);
        if (fread(q.data(), 80, n, pf) != (size_t)n) return 1;
        fclose(pf);
    } else q.resize(10 * (size_t)n);
    const double R = 1.25 * 100000 / (2 * M_PI);
    unsigned s = 12345;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0 / 16777216.0); };
    for (int i = 0; i < n && !pf; i++) {
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
        else if (mode.rfind("keys:", 0) == 0) {                                     // the library's order: buckets of the PREVIOUS search's length (file of n int32), longest first
            std::vector<int> key(n);
            FILE *kf = fopen(mode.c_str() + 5, "rb");
            if (!kf || fread(key.data(), 4, n, kf) != (size_t)n) { printf("no keys\n"); return 1; }
            fclose(kf);
            const int bw = getenv("PB_BW") ? atoi(getenv("PB_BW")) : 8, nb = 96 / bw;      // bucket width (the library: 8 -> 12 buckets from 40 up)
            auto bucket = [&](int it) { const int b = (it - 40) / bw; return it < 40 ? 0 : (b >= nb ? nb - 1 : b); };
            for (int b = nb - 1; b >= 0; b--) for (int i = 0; i < n; i++) if (bucket(key[i]) == b) perm.push_back(i);
        } else if (mode.rfind("groups:", 0) == 0 || mode == "groups_true") {       // wavefront-sized groups of consecutive plans (spatially coherent), the groups longest first
            std::vector<int> key(n);
            if (mode == "groups_true") key = it;
            else { FILE *kf = fopen(mode.c_str() + 7, "rb"); if (!kf || fread(key.data(), 4, n, kf) != (size_t)n) { printf("no keys\n"); return 1; } fclose(kf); }
            const int G = getenv("PB_GROUP") ? atoi(getenv("PB_GROUP")) : 64, ng = (n + G - 1) / G;
            std::vector<int> gk(ng, 0), go(ng);
            for (int i = 0; i < n; i++) gk[i / G] = std::max(gk[i / G], key[i]);
            for (int g = 0; g < ng; g++) go[g] = g;
            std::stable_sort(go.begin(), go.end(), [&](int a, int b) { return gk[a] > gk[b]; });
            for (int g : go) for (int i = g * G; i < std::min(n, (g + 1) * G); i++) perm.push_back(i);
        } else if (mode == "fold") {                                                  // longest first; the workgroups of the second round shortest first
            const int first = std::min(n, 256 * 256);
            for (int i = 0; i < first; i++) perm.push_back(order[n - 1 - i]);
            for (int i = 0; i < n - first; i++) perm.push_back(order[i]);
        } else {                                                                      // lpt: [short half | long | short half]
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
    {   // per wavefront: lane 0 holds the cycles, lane 1 the start (cycles since the first wave's start), lane 2 the hardware id
        struct W { double cyc, start; unsigned hw; int iters; };
        std::vector<W> w;
        double t0 = 1e300;
        for (int i = 0; i + 2 < n; i += 64) t0 = std::min(t0, l[i + 1]);
        for (int i = 0; i + 2 < n; i += 64) w.push_back({l[i], l[i + 1] - t0, (unsigned)l[i + 2], it[i]});
        { double c = 0, wl = 0, w0 = 1e300, w1 = 0; for (int i = 0; i + 4 < n; i += 64) { c += l[i]; wl += l[i + 3]; w0 = std::min(w0, l[i + 4]); w1 = std::max(w1, l[i + 4] + l[i + 3]); }
          printf("  core clock while the waves ran: %.3f GHz (cycles / 100-MHz ticks); first wave start to last wave end %.1f us\n", c / wl * 0.1, (w1 - w0) * 0.01); }
        double cy = 0, itw = 0, mxc = 0; for (auto &x : w) { cy += x.cyc; itw += x.iters; mxc = std::max(mxc, x.start + x.cyc); }
        printf("  waves %zu: mean %.0f cycles, %.1f wave-iterations (%.0f cycles each); last wave ends at %.0f cycles\n", w.size(), cy / w.size(), itw / w.size(), cy / itw, mxc);
        std::map<unsigned, std::vector<int>> by;                                   // SIMD -> its waves
        for (size_t k = 0; k < w.size(); k++) by[w[k].hw & 0xfffffff0u] .push_back((int)k);
        int lone = 0, pair = 0, more = 0; double cl = 0, il = 0, cp = 0, ip = 0;
        for (auto &kv : by) { if (kv.second.size() == 1) lone++; else if (kv.second.size() == 2) pair++; else more++;
            for (int k : kv.second) { if (kv.second.size() == 1) { cl += w[k].cyc; il += w[k].iters; } else { cp += w[k].cyc; ip += w[k].iters; } } }
        printf("  SIMDs with 1 / 2 / more waves: %d / %d / %d; cycles per iteration alone %.0f, sharing %.0f\n", lone, pair, more, cl / std::max(il, 1.0), cp / std::max(ip, 1.0));
        // wall-clock view (100-MHz ticks are global; the cycle counters of different XCDs are not aligned)
        std::vector<double> ws, wd; double w0 = 1e300;
        for (int i = 0; i + 4 < n; i += 64) { ws.push_back(l[i + 4]); wd.push_back(l[i + 3]); w0 = std::min(w0, l[i + 4]); }
        int late[6] = {0, 0, 0, 0, 0, 0};
        for (size_t k = 0; k < ws.size(); k++) { const double us = (ws[k] - w0) * 0.01; late[us < 5 ? 0 : us < 50 ? 1 : us < 200 ? 2 : us < 400 ? 3 : us < 600 ? 4 : 5]++; }
        printf("  wave starts: <5 us %d, <50 us %d, <200 us %d, <400 us %d, <600 us %d, later %d\n", late[0], late[1], late[2], late[3], late[4], late[5]);
        std::vector<int> ord(ws.size()); for (size_t k = 0; k < ws.size(); k++) ord[k] = (int)k;
        std::sort(ord.begin(), ord.end(), [&](int a, int b) { return ws[a] + wd[a] > ws[b] + wd[b]; });
        for (int k = 0; k < 10; k++) { const int j = ord[k]; printf("    wave %4d (block %d): start %.1f us, runs %.1f us, %d iterations (%.2f us each)\n", j, j / 4, (ws[j] - w0) * 0.01, wd[j] * 0.01, w[j].iters, wd[j] * 0.01 / w[j].iters); }
    }
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
