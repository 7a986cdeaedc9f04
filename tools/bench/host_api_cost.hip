// tools/bench/host_api_cost.hip -- what the HOST pays per HIP call while it enqueues a resident step (an SCA_NBR_AUTO step at N = 4096 is bound by
// the host's enqueue rate: tools/gpu/exp_host.py).  Each call 20 000 times back to back on otherwise idle streams, microseconds per call.
#include <hip/hip_runtime.h>
#include <chrono>
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
    run("hipMemcpyAsync D2H 4 B to pinned", [&](int) { (void)hipMemcpyAsync(hp, dp, 4, hipMemcpyDeviceToHost, A); }, true);
    printf("}}\n");
    return 0;
}
