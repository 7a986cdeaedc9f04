// tools/bench/host_api_cost.hip -- what the HOST pays per HIP call while it enqueues a resident step (an SCA_NBR_AUTO step at N = 4096 is bound by
// the host's enqueue rate: tools/gpu/exp_host.py).  Each call 20 000 times back to back on otherwise idle streams, microseconds per call.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <algorithm>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_empty(int *p) { if (p && threadIdx.x == 1 << 20) *p = 0; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t A, B;
    CHECK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    hipEvent_t e[4];
    for (auto &x : e) CHECK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
    unsigned *flag; CHECK(hipMalloc(&flag, 4)); CHECK(hipMemset(flag, 0, 4));
    int *hp; CHECK(hipHostMalloc(&hp, 4)); int *dp; CHECK(hipMalloc(&dp, 4));
    const int N = 20000;
    printf("{\"calls\": %d, \"us_per_call\": {", N);
    auto run = [&](const char *name, auto f, bool last = false) {
        hipDeviceSynchronize();
        for (int i = 0; i < 200; i++) f(i);
        hipDeviceSynchronize();
        const double t0 = now();
        for (int i = 0; i < N; i++) f(i);
        const double t1 = now();
        hipDeviceSynchronize();
        printf("\"%s\": %.2f%s", name, (t1 - t0) / N * 1e6, last ? "" : ", ");
        fflush(stdout);
    };
    run("hipLaunchKernelGGL (empty kernel, one stream)", [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, A, (int *)nullptr); });
    run("hipLaunchKernelGGL alternating two streams", [&](int i) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, (i & 1) ? A : B, (int *)nullptr); });
    run("hipEventRecord", [&](int i) { (void)hipEventRecord(e[i & 3], A); });
    run("hipEventRecord + hipStreamWaitEvent on the other stream", [&](int i) { (void)hipEventRecord(e[i & 3], A); (void)hipStreamWaitEvent(B, e[i & 3], 0); });
    run("kernel on A, record, wait on B, kernel on B", [&](int i) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, A, (int *)nullptr); (void)hipEventRecord(e[i & 3], A); (void)hipStreamWaitEvent(B, e[i & 3], 0); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, B, (int *)nullptr); });
    run("hipEventQuery (fired)", [&](int i) { (void)hipEventQuery(e[i & 3]); });
    run("hipStreamWaitValue32 (already satisfied)", [&](int) { (void)hipStreamWaitValue32(A, flag, 0u, hipStreamWaitValueEq, 1u); });
    run("kernel + hipStreamWaitValue32 + kernel", [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, A, (int *)nullptr); (void)hipStreamWaitValue32(A, flag, 0u, hipStreamWaitValueEq, 1u); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, A, (int *)nullptr); });
    run("hipMemcpyAsync D2H 4 B to pinned", [&](int) { (void)hipMemcpyAsync(hp, dp, 4, hipMemcpyDeviceToHost, A); });
    // two host threads, a stream each, at once: does the runtime let them enqueue side by side?  (per call, the slower thread)
    {
        hipDeviceSynchronize();
        std::atomic<int> go{0};
        double dt[2] = {0, 0};
        auto body = [&](int who) {
            hipStream_t s = who ? B : A;
            for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (int *)nullptr);
            go.fetch_add(1);
            while (go.load() < 2) {}
            const double t0 = now();
            for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (int *)nullptr);
            dt[who] = now() - t0;
        };
        std::thread th(body, 1);
        body(0);
        th.join();
        hipDeviceSynchronize();
        printf("\"hipLaunchKernelGGL, two host threads with a stream each (per call per thread)\": %.2f, ", std::max(dt[0], dt[1]) / N * 1e6);
    }
    {   // ... and the hand-over: a spinning helper thread is given {wait for an event of A on B, three launches on B, one record} while the caller
        // goes on with three launches on A and then waits for the helper: per round, against the same calls by the caller alone
        hipDeviceSynchronize();
        std::atomic<int> job{0}, done{0};
        std::atomic<bool> stop{false};
        std::thread th([&] {
            int seen = 0;
            while (!stop.load(std::memory_order_relaxed)) {
                const int j = job.load(std::memory_order_acquire);
                if (j == seen) continue;
                seen = j;
                (void)hipStreamWaitEvent(B, e[j & 3], 0);
                for (int k = 0; k < 3; k++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, B, (int *)nullptr);
                (void)hipEventRecord(e[2 + (j & 1)], B);
                done.store(j, std::memory_order_release);
            }
        });
        const int R = 5000;
        double t0 = now();
        for (int r = 1; r <= R; r++) {
            (void)hipEventRecord(e[r & 1], A);
            (void)hipStreamWaitEvent(B, e[r & 1], 0);
            for (int k = 0; k < 3; k++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, B, (int *)nullptr);
            (void)hipEventRecord(e[2 + (r & 1)], B);
            for (int k = 0; k < 3; k++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, A, (int *)nullptr);
        }
        const double alone = (now() - t0) / R * 1e6;
        hipDeviceSynchronize();
        t0 = now();
        for (int r = 1; r <= R; r++) {
            (void)hipEventRecord(e[r & 1], A);
            job.store(r, std::memory_order_release);
            for (int k = 0; k < 3; k++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, A, (int *)nullptr);
            while (done.load(std::memory_order_acquire) != r) {}
        }
        const double helped = (now() - t0) / R * 1e6;
        stop.store(true);
        th.join();
        hipDeviceSynchronize();
        printf("\"round {record A | wait+3 launches+record on B | 3 launches on A}: caller alone\": %.2f, \"... with a spinning helper thread for the B part\": %.2f", alone, helped);
    }
    printf("}}\n");
    return 0;
}
