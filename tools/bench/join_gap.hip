// tools/bench/join_gap.hip -- what a cross-stream JOIN costs on the critical path when the awaited work finished long ago.
// Stream A: kernel L (long, ~100 us) then kernel C; stream B: kernel S (short, ~20 us) started beside L.  C must run after S.
//   0  no dependency at all (the floor: a dependent launch on one stream)
//   1  hipEventRecord(e, B) ... hipStreamWaitEvent(A, e) between L and C
//   2  S writes a flag; hipStreamWaitValue32(A, flag, seq, GEQ) between L and C
//   3  no stream operation: C's first wavefront polls the flag S wrote (a device-side join)
// The gap = C's first timestamp - L's last timestamp (100-MHz wall clock read inside the kernels), median over 200 repetitions with the
// host enqueueing far ahead.   hipcc --offload-arch=gfx950 -O3 -o /tmp/join_gap tools/bench/join_gap.hip && /tmp/join_gap
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_busy(unsigned long long *stamp, int slot, long long ticks, unsigned *flag, unsigned seq) {
    const unsigned long long t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < ticks) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamp[slot] = wall_clock64(); if (flag) { __threadfence(); atomicExch(flag, seq); } }
}
__global__ void k_consumer(unsigned long long *stamp, int slot, const unsigned *flag, unsigned seq) {
    if (flag) { while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < seq) __builtin_amdgcn_s_sleep(1); }
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[slot] = wall_clock64();
}

int main() {
    hipStream_t A, B;
    CHECK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    hipEvent_t fork, join;
    CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    const int reps = 200;
    unsigned long long *stamp; unsigned *flag;
    CHECK(hipMalloc(&stamp, sizeof(unsigned long long) * 2 * reps)); CHECK(hipMalloc(&flag, sizeof(unsigned)));
    std::vector<unsigned long long> h(2 * reps);
    printf("{\"what\": \"gap between the end of a 100-us kernel on stream A and the start of the next kernel on A, which depends on a 20-us kernel that ran beside it on stream B\", \"rows\": [\n");
    const char *names[] = {"no dependency (one stream)", "hipStreamWaitEvent on the other stream's event", "hipStreamWaitValue32 on a flag the other stream's kernel wrote",
                           "no stream operation: the consumer kernel polls the flag"};
    for (int variant = 0; variant < 4; variant++) {
        CHECK(hipMemset(flag, 0, sizeof(unsigned)));
        CHECK(hipDeviceSynchronize());
        for (int r = 0; r < reps; r++) {
            const unsigned seq = (unsigned)(r + 1);
            CHECK(hipEventRecord(fork, A));
            CHECK(hipStreamWaitEvent(B, fork, 0));
            hipLaunchKernelGGL(k_busy, dim3(64), dim3(64), 0, B, stamp, 0, 2000LL, flag, seq);                 // S: 20 us, beside L
            if (variant == 1) CHECK(hipEventRecord(join, B));
            hipLaunchKernelGGL(k_busy, dim3(64), dim3(64), 0, A, stamp, 2 * r, 10000LL, (unsigned *)nullptr, 0u);   // L: 100 us
            if (variant == 1) CHECK(hipStreamWaitEvent(A, join, 0));
            if (variant == 2) CHECK(hipStreamWaitValue32(A, flag, seq, hipStreamWaitValueGte, 0xffffffffu));
            hipLaunchKernelGGL(k_consumer, dim3(64), dim3(64), 0, A, stamp, 2 * r + 1, variant == 3 ? flag : (const unsigned *)nullptr, seq);
        }
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), stamp, sizeof(unsigned long long) * 2 * reps, hipMemcpyDeviceToHost));
        std::vector<double> gaps;
        for (int r = 20; r < reps; r++) gaps.push_back((double)((long long)h[2 * r + 1] - (long long)h[2 * r]) / 100.0);
        std::sort(gaps.begin(), gaps.end());
        printf("  {\"join\": \"%s\", \"gap_us_median\": %.2f, \"gap_us_p10\": %.2f, \"gap_us_p90\": %.2f}%s\n", names[variant], gaps[gaps.size() / 2], gaps[gaps.size() / 10],
               gaps[gaps.size() * 9 / 10], variant < 3 ? "," : "");
    }
    printf("]}\n");
    return 0;
}
