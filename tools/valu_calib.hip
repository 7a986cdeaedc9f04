// tools/valu_calib.hip -- what a gfx950 SIMD issues per microsecond, per instruction class, with 1 / 2 / 4 wavefronts per SIMD:
// the calibration of bench.py's `valu_issue_frac` (round 4 priced every VALU wave-instruction at 4 cycles; MI355X_MICROARCH.md says
// 2 cycles for 32-bit VALU on the SIMD-32, 4 for ONE wave's own stream -- which of the two applies to k_replan's mix had never been
// measured the way FETCH_SIZE was).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_calib tools/valu_calib.hip && /tmp/valu_calib > gpurun_out/valu_calib.json
//
// Per class: a kernel whose loop body is 64 instructions of that class on 8 independent register chains (ILP 8: issue-bound, not
// latency-bound) -- and the same with ONE chain (ILP 1: the dependent-issue latency).  256 x w workgroups of 256 threads = w wavefronts
// on every SIMD of the chip (w = 1, 2, 4).  Reported: wave-instructions per microsecond per SIMD, and the cycles per instruction that
// is at the shader clock measured beside it (a chain of dependent s_add over a wall_clock64 interval).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// eight independent chains r0..r7 (doubles) / i0..i7 (ints); OP8 issues the class's instruction once per chain
#define REP8(X) X X X X X X X X

#define DEF_KERNEL_F64(NAME, ASM8, ASM1)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(double *out, int trips, int ilp1) {                                                \
        double r0 = threadIdx.x * 1e-3 + 1.0, r1 = r0 + 0.1, r2 = r0 + 0.2, r3 = r0 + 0.3, r4 = r0 + 0.4, r5 = r0 + 0.5, r6 = r0 + 0.6, r7 = r0 + 0.7; \
        const double b = 1.0000001, c = 1e-9;                                                                                     \
        if (ilp1) { for (int t = 0; t < trips; t++) { asm volatile(REP8(REP8(ASM1)) : "+v"(r0) : "v"(b), "v"(c) : "vcc", "s20", "s21"); } }               \
        else { for (int t = 0; t < trips; t++) { asm volatile(REP8(ASM8) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b), "v"(c) : "vcc", "s20", "s21"); } } \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                                        \
    }
#define DEF_KERNEL_I32(NAME, ASM8, ASM1)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(double *out, int trips, int ilp1) {                                                \
        int r0 = threadIdx.x + 1, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;       \
        const int b = 3, c = 0x55aa;                                                                                              \
        if (ilp1) { for (int t = 0; t < trips; t++) { asm volatile(REP8(REP8(ASM1)) : "+v"(r0) : "v"(b), "v"(c) : "vcc", "s20", "s21"); } }               \
        else { for (int t = 0; t < trips; t++) { asm volatile(REP8(ASM8) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b), "v"(c) : "vcc", "s20", "s21"); } } \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7);                              \
    }
// operands: %0..%7 the chains, %8 = b, %9 = c   (one chain: %0, %1 = b, %2 = c)
#define A8(op, rest) op " %0, %0" rest "\n" op " %1, %1" rest "\n" op " %2, %2" rest "\n" op " %3, %3" rest "\n" op " %4, %4" rest "\n" op " %5, %5" rest "\n" op " %6, %6" rest "\n" op " %7, %7" rest "\n"
#define A1(op, rest) op " %0, %0" rest "\n"

DEF_KERNEL_F64(k_fma_f64, A8("v_fma_f64", ", %8, %9"), A1("v_fma_f64", ", %1, %2"))
DEF_KERNEL_F64(k_mul_f64, A8("v_mul_f64", ", %8"), A1("v_mul_f64", ", %1"))
DEF_KERNEL_F64(k_add_f64, A8("v_add_f64", ", %9"), A1("v_add_f64", ", %2"))
DEF_KERNEL_F64(k_rcp_f64, "v_rcp_f64 %0, %0\nv_rcp_f64 %1, %1\nv_rcp_f64 %2, %2\nv_rcp_f64 %3, %3\nv_rcp_f64 %4, %4\nv_rcp_f64 %5, %5\nv_rcp_f64 %6, %6\nv_rcp_f64 %7, %7\n", "v_rcp_f64 %0, %0\n")
DEF_KERNEL_F64(k_rsq_f64, "v_rsq_f64 %0, %0\nv_rsq_f64 %1, %1\nv_rsq_f64 %2, %2\nv_rsq_f64 %3, %3\nv_rsq_f64 %4, %4\nv_rsq_f64 %5, %5\nv_rsq_f64 %6, %6\nv_rsq_f64 %7, %7\n", "v_rsq_f64 %0, %0\n")
// compare into vcc + select: the pair the branch-free libm restatement is made of (two 32-bit selects per double)
DEF_KERNEL_F64(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %8\nv_cmp_lt_f64 vcc, %1, %8\nv_cmp_lt_f64 vcc, %2, %8\nv_cmp_lt_f64 vcc, %3, %8\nv_cmp_lt_f64 vcc, %4, %8\nv_cmp_lt_f64 vcc, %5, %8\nv_cmp_lt_f64 vcc, %6, %8\nv_cmp_lt_f64 vcc, %7, %8\n", "v_cmp_lt_f64 vcc, %0, %1\n")
// (a bare stream of v_cndmask_b32 ... vcc with nobody writing vcc measured 22.7 cycles per instruction at any occupancy -- not what code
// issues: a select follows the compare that made its mask.  So: the compare alone, and the compare + select pair.)
DEF_KERNEL_I32(k_cmp_u32, "v_cmp_lt_u32 vcc, %0, %8\nv_cmp_lt_u32 vcc, %1, %8\nv_cmp_lt_u32 vcc, %2, %8\nv_cmp_lt_u32 vcc, %3, %8\nv_cmp_lt_u32 vcc, %4, %8\nv_cmp_lt_u32 vcc, %5, %8\nv_cmp_lt_u32 vcc, %6, %8\nv_cmp_lt_u32 vcc, %7, %8\n", "v_cmp_lt_u32 vcc, %0, %1\n")
DEF_KERNEL_I32(k_cmp_cndmask, "v_cmp_lt_u32 vcc, %0, %9\nv_cndmask_b32 %0, %0, %8, vcc\nv_cmp_lt_u32 vcc, %1, %9\nv_cndmask_b32 %1, %1, %8, vcc\nv_cmp_lt_u32 vcc, %2, %9\nv_cndmask_b32 %2, %2, %8, vcc\nv_cmp_lt_u32 vcc, %3, %9\nv_cndmask_b32 %3, %3, %8, vcc\n", "v_cmp_lt_u32 vcc, %0, %2\nv_cndmask_b32 %0, %0, %1, vcc\n")
DEF_KERNEL_I32(k_cndmask_sgpr, "v_cndmask_b32 %0, %0, %8, s[20:21]\nv_cndmask_b32 %1, %1, %8, s[20:21]\nv_cndmask_b32 %2, %2, %8, s[20:21]\nv_cndmask_b32 %3, %3, %8, s[20:21]\nv_cndmask_b32 %4, %4, %8, s[20:21]\nv_cndmask_b32 %5, %5, %8, s[20:21]\nv_cndmask_b32 %6, %6, %8, s[20:21]\nv_cndmask_b32 %7, %7, %8, s[20:21]\n", "v_cndmask_b32 %0, %0, %1, s[20:21]\n")
DEF_KERNEL_I32(k_mov_b32, "v_mov_b32 %0, %8\nv_mov_b32 %1, %8\nv_mov_b32 %2, %8\nv_mov_b32 %3, %8\nv_mov_b32 %4, %8\nv_mov_b32 %5, %8\nv_mov_b32 %6, %8\nv_mov_b32 %7, %8\n", "v_mov_b32 %0, %1\n")
DEF_KERNEL_I32(k_add_u32, A8("v_add_u32", ", %8"), A1("v_add_u32", ", %1"))
DEF_KERNEL_I32(k_and_b32, A8("v_and_b32", ", %9"), A1("v_and_b32", ", %2"))
DEF_KERNEL_I32(k_lshl_b32, "v_lshlrev_b32 %0, 1, %0\nv_lshlrev_b32 %1, 1, %1\nv_lshlrev_b32 %2, 1, %2\nv_lshlrev_b32 %3, 1, %3\nv_lshlrev_b32 %4, 1, %4\nv_lshlrev_b32 %5, 1, %5\nv_lshlrev_b32 %6, 1, %6\nv_lshlrev_b32 %7, 1, %7\n", "v_lshlrev_b32 %0, 1, %0\n")
DEF_KERNEL_I32(k_mul_lo_u32, A8("v_mul_lo_u32", ", %8"), A1("v_mul_lo_u32", ", %1"))
DEF_KERNEL_I32(k_fma_f32, A8("v_fma_f32", ", %8, %9"), A1("v_fma_f32", ", %1, %2"))
DEF_KERNEL_F64(k_lshl_b64, "v_lshlrev_b64 %0, 1, %0\nv_lshlrev_b64 %1, 1, %1\nv_lshlrev_b64 %2, 1, %2\nv_lshlrev_b64 %3, 1, %3\nv_lshlrev_b64 %4, 1, %4\nv_lshlrev_b64 %5, 1, %5\nv_lshlrev_b64 %6, 1, %6\nv_lshlrev_b64 %7, 1, %7\n", "v_lshlrev_b64 %0, 1, %0\n")

// LDS table reads (the restated libm's tables live in LDS): ds_read_b64 at lane-dependent addresses, 8 in flight
__global__ __launch_bounds__(256) void k_ds_read_b64(double *out, int trips, int ilp1) {
    __shared__ double tab[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) tab[i] = i;
    __syncthreads();
    double acc = 0.0;
    int idx = (threadIdx.x * 37) & 2047;
    for (int t = 0; t < trips; t++) {
#pragma unroll
        for (int k = 0; k < 64; k++) { acc += tab[(idx + k * 33) & 2047]; }
        idx = (idx + 7) & 2047;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + ilp1;
}

// shader clock: dependent s_add_u32 chain (1 SALU op per cycle... issue 4 cycles apart for one wave) is not a clock; use the cycle counter
// against the 100-MHz wall clock instead
__global__ void k_clock(unsigned long long *out) {
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    unsigned long long w1 = w0;
    while (w1 - w0 < 200000ull) w1 = wall_clock64();                     // 2 ms of the 100-MHz counter
    const unsigned long long c1 = __builtin_readcyclecounter();
    out[0] = w1 - w0; out[1] = c1 - c0;
}

typedef void (*kern_t)(double *, int, int);
struct Class { const char *name; kern_t k; const char *what; };

int main() {
    double *out;
    CHECK(hipMalloc(&out, sizeof(double) * 256 * 256 * 8));
    unsigned long long *clk, hclk[2];
    CHECK(hipMalloc(&clk, 16));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, simds = cus * 4;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const Class cls[] = {
        {"v_fma_f64", k_fma_f64, "fp64 fused multiply-add"}, {"v_mul_f64", k_mul_f64, "fp64 multiply"}, {"v_add_f64", k_add_f64, "fp64 add"},
        {"v_rcp_f64", k_rcp_f64, "fp64 reciprocal seed (transcendental unit)"}, {"v_rsq_f64", k_rsq_f64, "fp64 reciprocal square root seed"},
        {"v_cmp_lt_f64", k_cmp_f64, "fp64 compare into vcc"}, {"v_cmp_lt_u32", k_cmp_u32, "32-bit compare into vcc"}, {"v_cmp_lt_u32+v_cndmask_b32", k_cmp_cndmask, "compare + the select that reads its mask, counted as TWO instructions"},
        {"v_cndmask_b32(sgpr mask)", k_cndmask_sgpr, "32-bit select on a mask held in an SGPR pair (a double select is two)"},
        {"v_mov_b32", k_mov_b32, "32-bit move"}, {"v_add_u32", k_add_u32, "32-bit integer add"}, {"v_and_b32", k_and_b32, "32-bit logic"},
        {"v_lshlrev_b32", k_lshl_b32, "32-bit shift"}, {"v_lshlrev_b64", k_lshl_b64, "64-bit shift"}, {"v_mul_lo_u32", k_mul_lo_u32, "32-bit integer multiply"},
        {"v_fma_f32", k_fma_f32, "fp32 fused multiply-add (reference point: 2 cycles on the SIMD-32)"},
        {"ds_read_b64", k_ds_read_b64, "LDS read of a double at a lane-dependent address + the add that consumes it"},
    };
    // clock
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(1), 0, 0, clk);
    CHECK(hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost));
    // the cycle counter of gfx9 (s_memtime) runs at a constant 100 MHz on this family; if it does, cycles cannot be derived from it
    const double counter_mhz = (double)hclk[1] / ((double)hclk[0] / 100.0);
    printf("{\n \"device\": \"%s\", \"cus\": %d, \"simds\": %d, \"clockRate_khz_reported\": %d, \"cycle_counter_mhz\": %.1f,\n \"classes\": [\n", prop.gcnArchName, cus, simds, prop.clockRate, counter_mhz);
    const double mhz = prop.clockRate / 1000.0;                          // the reported peak shader clock; the rates below are what was measured
    const int ncls = (int)(sizeof(cls) / sizeof(cls[0]));
    for (int ci = 0; ci < ncls; ci++) {
        printf("  {\"op\": \"%s\", \"what\": \"%s\"", cls[ci].name, cls[ci].what);
        for (int ilp1 = 0; ilp1 < 2; ilp1++) {
            for (int w : {1, 2, 4}) {
                if (ilp1 && w != 1) continue;
                const int trips = 4000;
                const dim3 grid(cus * w), block(256);
                hipLaunchKernelGGL(cls[ci].k, grid, block, 0, 0, out, 200, ilp1);          // warm-up
                CHECK(hipDeviceSynchronize());
                float best = 1e30f;
                for (int rep = 0; rep < 3; rep++) {
                    CHECK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(cls[ci].k, grid, block, 0, 0, out, trips, ilp1);
                    CHECK(hipEventRecord(e1, 0));
                    CHECK(hipEventSynchronize(e1));
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                const double wave_insts = (double)trips * 64.0 * (double)(cus * w) * 4.0;      // per kernel: trips x 64 per wave x waves
                const double per_us_per_simd = wave_insts / (best * 1e3) / simds;
                if (ilp1) printf(", \"dependent_chain_cycles_at_reported_clock\": %.2f", mhz / per_us_per_simd);
                else printf(", \"w%d\": {\"wave_insts_per_us_per_simd\": %.1f, \"cycles_per_inst_at_reported_clock\": %.2f}", w, per_us_per_simd, mhz / per_us_per_simd);
            }
        }
        printf("}%s\n", ci + 1 < ncls ? "," : "");
    }
    printf(" ]\n}\n");
    return 0;
}
