// Does a hipGraph shorten the gap between DEPENDENT short kernels on this stack?  (The step of a small scene is a chain of 13-17 kernels of
// 4-40 us; the host enqueues far ahead, so only the device-side gap between a kernel's end and its successor's start is left to win.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/graph_gap tools/bench/graph_gap.hip && /tmp/graph_gap
// Prints the time per kernel of a chain of 16 dependent kernels (each ~BUSY us of dependent FMAs in one wavefront per CU), launched
// (a) one by one into a stream, the host running ahead, (b) as one captured graph per chain.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_busy(double *p, int iters) {
    double a = p[threadIdx.x & 63];
    for (int i = 0; i < iters; i++) a = __builtin_fma(a, 1.0000001, 1e-9);
    if (a == 12345.678) p[0] = a;
}

int main(int argc, char **argv) {
    const int chain = 16, reps = 400;
    double *p;
    CK(hipMalloc(&p, 4096));
    CK(hipMemset(p, 0, 4096));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int iters : {0, 400, 2000}) {
        auto run_stream = [&]() { for (int k = 0; k < chain; k++) hipLaunchKernelGGL(k_busy, dim3(256), dim3(64), 0, s, p, iters); };
        for (int w = 0; w < 20; w++) run_stream();
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) run_stream();
        CK(hipStreamSynchronize(s));
        const double us_stream = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * chain);
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        run_stream();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 20; w++) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        const double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * chain);
        // one kernel alone, timed by events over many back-to-back independent launches is not the point: the chain's per-kernel time IS gap + kernel
        printf("iters %5d: per dependent kernel %.2f us as stream launches, %.2f us inside a graph of %d\n", iters, us_stream, us_graph, chain);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
