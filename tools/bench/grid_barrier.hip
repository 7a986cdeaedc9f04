// What does a device-wide barrier cost on MI355X, against the gap between two dependent dispatches on one queue?
// (the c3 step is seven dependent kernels with 35 us of work and 41 us of idle chip between them.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/grid_barrier tools/bench/grid_barrier.hip && /tmp/grid_barrier
//
// One persistent launch of G workgroups of 256 threads (all resident: G <= 2048 at <= 64 VGPRs), R rounds of
//     [a little work: every workgroup writes one word per round, reads its neighbour's word of the PREVIOUS round and checks it] + barrier
// in three barrier forms:
//   flat   one agent-scope counter: arrive = atomic add (release), wait = spin on an atomic load (acquire) until G * round arrivals
//   xcd    two levels: a counter per XCD (workgroup id mod 8 = the XCD the dispatcher puts it on), the last arriver of an XCD adds to the
//          global counter; everybody spins on the global one
//   tree   like xcd, but the waiters spin on a per-XCD release word that the XCD's last arriver sets once the global count is complete
//          (one spinner per XCD on the contended global word)
// against the same work as R dependent launches of G workgroups on one stream (and inside one hipGraph).
// Reports us per round.  The neighbour check proves the barrier orders the writes (a count of violations is printed; it must be 0).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Bar {
    unsigned *global;        // [1]
    unsigned *xcd;           // [8 * 32] one counter per XCD, 128 B apart
    unsigned *release;       // [8 * 32] per-XCD release word (tree form)
};

__device__ __forceinline__ unsigned ld_acq(unsigned *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add_rel(unsigned *p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_rel(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

template <int FORM>
__device__ __forceinline__ void grid_barrier(const Bar &B, unsigned G, unsigned round /* 1, 2, ... */) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FORM == 0) {
            add_rel(B.global, 1u);
            while (ld_acq(B.global) < G * round) __builtin_amdgcn_s_sleep(1);
        } else {
            const unsigned x = blockIdx.x & 7u;
            const unsigned in_xcd = (G >> 3) + ((G & 7u) > x ? 1u : 0u);                  // workgroups with this id mod 8
            const unsigned prev = add_rel(B.xcd + 32 * x, 1u);
            const bool last = prev + 1 == in_xcd * round;
            if (FORM == 1) {
                if (last) add_rel(B.global, 1u);
                const unsigned want = (G < 8u ? G : 8u) * round;
                while (ld_acq(B.global) < want) __builtin_amdgcn_s_sleep(1);
            } else {
                if (last) {
                    add_rel(B.global, 1u);
                    const unsigned want = (G < 8u ? G : 8u) * round;
                    while (ld_acq(B.global) < want) __builtin_amdgcn_s_sleep(1);
                    st_rel(B.release + 32 * x, round);
                } else
                    while (ld_acq(B.release + 32 * x) < round) __builtin_amdgcn_s_sleep(1);
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void round_work(unsigned *words, unsigned *bad, unsigned G, unsigned r) {
    // check the neighbour's word of the previous round (written before the barrier that ended it), then write this round's
    if (threadIdx.x == 0) {
        const unsigned nb = (blockIdx.x + 1) % G;
        if (r > 1 && __hip_atomic_load(words + ((r - 1) & 1u) * 32 * 4096 + 32 * nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != r - 1) atomicAdd(bad, 1u);
    }
}
__device__ __forceinline__ void round_write(unsigned *words, unsigned r) {
    if (threadIdx.x == 0) __hip_atomic_store(words + 32 * blockIdx.x, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int FORM>
__global__ __launch_bounds__(256) void k_persistent(Bar B, unsigned *words, unsigned *bad, unsigned G, unsigned rounds, unsigned base) {
    for (unsigned r = 1; r <= rounds; r++) {
        round_work(words, bad, G, r);
        // (everybody must have read round r - 1 before anybody overwrites it: the write goes behind a barrier of its own in a real
        // pipeline -- here the read and the write are separated by the barrier pair of two consecutive rounds, so write first round's
        // value only after the check: two barriers per round would double the cost being measured; instead the words are double-buffered)
        round_write(words + (r & 1u) * 32 * 4096, r);
        grid_barrier<FORM>(B, G, base + r);
    }
}
// the same round as one launch
__global__ __launch_bounds__(256) void k_round(unsigned *words, unsigned *bad, unsigned G, unsigned r) {
    round_work(words, bad, G, r);
    round_write(words + (r & 1u) * 32 * 4096, r);
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Bar B;
    unsigned *words, *bad;
    CK(hipMalloc(&B.global, 256)); CK(hipMalloc(&B.xcd, 8 * 32 * 4)); CK(hipMalloc(&B.release, 8 * 32 * 4));
    CK(hipMalloc(&words, 2 * 32 * 4096 * 4)); CK(hipMalloc(&bad, 4));
    const unsigned rounds = 200;
    printf("%6s %12s %12s %12s %14s %14s\n", "G", "flat us", "xcd us", "tree us", "launches us", "graph us");
    for (unsigned G : {8u, 64u, 128u, 256u, 512u, 1024u, 1536u}) {
        double us[3];
        unsigned viol = 0;
        for (int form = 0; form < 3; form++) {
            CK(hipMemsetAsync(B.global, 0, 256, s)); CK(hipMemsetAsync(B.xcd, 0, 8 * 32 * 4, s)); CK(hipMemsetAsync(B.release, 0, 8 * 32 * 4, s));
            CK(hipMemsetAsync(words, 0, 2 * 32 * 4096 * 4, s)); CK(hipMemsetAsync(bad, 0, 4, s));
            auto launch = [&](unsigned base) {
                if (form == 0) hipLaunchKernelGGL(k_persistent<0>, dim3(G), dim3(256), 0, s, B, words, bad, G, rounds, base);
                else if (form == 1) hipLaunchKernelGGL(k_persistent<1>, dim3(G), dim3(256), 0, s, B, words, bad, G, rounds, base);
                else hipLaunchKernelGGL(k_persistent<2>, dim3(G), dim3(256), 0, s, B, words, bad, G, rounds, base);
            };
            // (the neighbour check compares with r - 1 of THIS launch: each launch starts from zeroed words, so reset between launches)
            launch(0);
            CK(hipStreamSynchronize(s));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int reps = 5;
            float ms_sum = 0;
            for (int k = 0; k < reps; k++) {
                CK(hipMemsetAsync(words, 0, 2 * 32 * 4096 * 4, s));
                CK(hipEventRecord(e0, s));
                launch((unsigned)(k + 1) * rounds);
                CK(hipEventRecord(e1, s));
                CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                ms_sum += ms;
            }
            us[form] = ms_sum / reps * 1e3 / rounds;
            unsigned b; CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
            viol += b;
            CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
        // the same rounds as dependent launches
        CK(hipMemsetAsync(words, 0, 2 * 32 * 4096 * 4, s)); CK(hipMemsetAsync(bad, 0, 4, s));
        auto chain = [&]() { for (unsigned r = 1; r <= rounds; r++) hipLaunchKernelGGL(k_round, dim3(G), dim3(256), 0, s, words, bad, G, r); };
        chain(); CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < 5; k++) chain();
        CK(hipStreamSynchronize(s));
        const double us_launch = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (5 * rounds);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        chain();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < 5; k++) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        const double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (5 * rounds);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        unsigned b2; CK(hipMemcpy(&b2, bad, 4, hipMemcpyDeviceToHost));
        printf("%6u %12.2f %12.2f %12.2f %14.2f %14.2f   ordering violations %u (+%u in the launch chain)\n", G, us[0], us[1], us[2], us_launch, us_graph, viol, b2);
    }
    return 0;
}
