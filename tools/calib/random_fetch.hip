// Calibration: how many scattered reads per second does the memory system deliver, by read size?
// Every lane reads RECORD bytes (a multiple of 16, aligned to RECORD) at a pseudo-random record index of a table
// far larger than the caches; lanes are independent (no dependent chains), 8 waves per SIMD.
// usage: random_fetch TABLE_MB   -> one line per record size: bytes, reads/s, useful GB/s
// With rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum this also calibrates those counters for scattered reads.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int QUADS>
__global__ __launch_bounds__(256) void k_fetch(const float4 *__restrict__ table, uint64_t n_records, uint32_t reads_per_lane,
                                               float *__restrict__ out) {
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.0f;
    for (uint32_t r = 0; r < reads_per_lane; ++r) {
        const uint64_t idx = mix64(gid * 0x9E3779B97F4A7C15ull + r) % n_records;
        const float4 *p = table + idx * QUADS;
#pragma unroll
        for (int q = 0; q < QUADS; ++q) {
            float4 v = p[q];
            acc += v.x + v.w;
        }
    }
    if (acc == 123.456f) out[0] = acc; // never true: keeps the loads alive
}

template <int QUADS>
static int run(const float4 *table, uint64_t table_bytes, float *out) {
    const uint64_t n_records = table_bytes / (16ull * QUADS);
    const uint32_t grid = 256 * 8 * 4, reads = 256;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_fetch<QUADS>, dim3(grid), dim3(256), 0, 0, table, n_records, 16u, out); // warm-up
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(k_fetch<QUADS>, dim3(grid), dim3(256), 0, 0, table, n_records, reads, out);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double n = (double)grid * 256 * reads;
    printf("record %4d B: %.3e reads/s, %.2f TB/s useful, %.3f ms\n", 16 * QUADS, n / (ms * 1e-3), n * 16 * QUADS / (ms * 1e-3) / 1e12, ms);
    return 0;
}

int main(int argc, char **argv) {
    const uint64_t mb = argc > 1 ? strtoull(argv[1], nullptr, 10) : 2048;
    const uint64_t bytes = mb << 20;
    float4 *table = nullptr;
    float *out = nullptr;
    CHECK(hipMalloc(&table, bytes));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(table, 0, bytes));
    if (run<2>(table, bytes, out) || run<4>(table, bytes, out) || run<8>(table, bytes, out) || run<16>(table, bytes, out)) return 1;
    return 0;
}
