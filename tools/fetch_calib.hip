// fetch_calib.hip -- known-bytes microbenchmark for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against the
// access shapes of this library (MI355X_MICROARCH.md, HBM section: FETCH_SIZE halves wide coalesced reads; other widths are
// uncalibrated).  Build: hipcc --offload-arch=gfx950 -O3 -o scratch/fetch_calib tools/fetch_calib.hip
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and once more with WRITE_SIZE); it prints the bytes each kernel must
// fetch from memory (tables far larger than the 256 MiB Infinity Cache, every line touched once).
//   k_stream16 : 16 B per lane, coalesced (the shape the guide calibrated: FETCH_SIZE reads 1/2)
//   k_stream8  : 8 B per lane, coalesced (kx / ky / kz arrays, neighbour distSq lists)
//   k_gather48 : one 48-byte record per lane at a random index, read as 3 x 16 B (k_solve / K1 reading PubRec of neighbours)
//   k_write48  : one 48-byte record per lane, consecutive (the moved records written by the integrate stage)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

struct alignas(16) Rec { double a, b, c; float d, e, f; unsigned g; double h; };
static_assert(sizeof(Rec) == 48, "");

__global__ void k_stream16(const double2 *p, size_t n, double *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const double2 v = p[i]; if (v.x == 1.2345 && v.y == 5.4321) out[0] = v.x; }
}
__global__ void k_stream8(const double *p, size_t n, double *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const double v = p[i]; if (v == 1.2345) out[0] = v; }
}
__global__ void k_gather48(const Rec *t, const unsigned *idx, size_t n, double *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const Rec r = t[idx[i]]; if (r.a == 1.2345 && r.h == 5.4321 && r.g == 77u) out[0] = r.b + r.d; }
}
__global__ void k_write48(Rec *t, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { Rec r; r.a = (double)i; r.b = 1.0; r.c = 2.0; r.d = 3.f; r.e = 4.f; r.f = 5.f; r.g = (unsigned)i; r.h = 0.5; t[i] = r; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); return 1; } } while (0)

int main() {
    const size_t bytes = (size_t)2 << 30;                          // 2 GiB tables
    void *buf = nullptr, *flush = nullptr; double *out = nullptr; unsigned *idx = nullptr;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&flush, bytes)); CK(hipMalloc((void **)&out, 64));
    CK(hipMemset(buf, 0, bytes)); CK(hipMemset(flush, 1, bytes));
    const size_t n_g = (size_t)1 << 20;                            // a million random records
    const size_t n_rec = bytes / sizeof(Rec);
    std::vector<unsigned> h(n_g);
    std::set<unsigned long long> lines64, lines128;
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < n_g; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        h[i] = (unsigned)(s % n_rec);
        const unsigned long long a = (unsigned long long)h[i] * 48, b = a + 47;
        for (unsigned long long l = a / 64; l <= b / 64; l++) lines64.insert(l);
        for (unsigned long long l = a / 128; l <= b / 128; l++) lines128.insert(l);
    }
    CK(hipMalloc((void **)&idx, n_g * 4)); CK(hipMemcpy(idx, h.data(), n_g * 4, hipMemcpyHostToDevice));
    auto evict = [&]() { return hipMemset(flush, 2, bytes); };    // 2 GiB of writes between measurements: nothing of buf stays cached
    const size_t n16 = bytes / 16, n8 = ((size_t)1 << 30) / 8;
    CK(evict()); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_stream16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (const double2 *)buf, n16, out);
    CK(hipDeviceSynchronize()); CK(evict()); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_stream8, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, 0, (const double *)buf, n8, out);
    CK(hipDeviceSynchronize()); CK(evict()); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_gather48, dim3((unsigned)((n_g + 255) / 256)), dim3(256), 0, 0, (const Rec *)buf, idx, n_g, out);
    CK(hipDeviceSynchronize()); CK(evict()); CK(hipDeviceSynchronize());
    const size_t n_w = ((size_t)1 << 30) / 48;
    hipLaunchKernelGGL(k_write48, dim3((unsigned)((n_w + 255) / 256)), dim3(256), 0, 0, (Rec *)buf, n_w);
    CK(hipDeviceSynchronize());
    std::printf("{\"k_stream16_bytes\": %zu, \"k_stream8_bytes\": %zu, \"k_gather48_bytes_touched_64B_lines\": %zu, "
                "\"k_gather48_bytes_touched_128B_lines\": %zu, \"k_gather48_index_bytes\": %zu, \"k_gather48_payload_bytes\": %zu, "
                "\"k_write48_bytes\": %zu}\n",
                n16 * 16, n8 * 8, lines64.size() * 64, lines128.size() * 128, n_g * 4, n_g * 48, n_w * 48);
    return 0;
}
