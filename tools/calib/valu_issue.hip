// Calibration: how many wave64 VALU instructions per second does the chip issue, by instruction kind and by waves per
// SIMD, and at what shader clock?  Every wave runs a long unrolled stream of independent instructions on private
// registers; the clock is the ratio of the shader cycle counter to the 100 MHz wall counter inside the same kernel.
// usage: valu_issue   -> one line per (kind, waves per SIMD): wave-instructions/s, measured clock, cycles per
// instruction per SIMD at that clock
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHECK(x)                                                                    \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            return 1;                                                               \
        }                                                                           \
    } while (0)

constexpr int kIters = 4096, kUnroll = 16;

struct Clocks {
    unsigned long long shader, wall;
};

#define KERNEL(NAME, ASM, ...)                                                                        \
    __global__ __launch_bounds__(256) void NAME(Clocks *out, float seed) {                            \
        float a[kUnroll];                                                                             \
        _Pragma("unroll") for (int i = 0; i < kUnroll; ++i) a[i] = seed + (float)(threadIdx.x + i);   \
        const float m = seed * 1.0001f, c = seed * 0.5f;                                              \
        (void)m;                                                                                      \
        (void)c;                                                                                      \
        const unsigned long long s0 = __builtin_readcyclecounter(), w0 = wall_clock64();              \
        for (int it = 0; it < kIters; ++it) {                                                         \
            _Pragma("unroll") for (int i = 0; i < kUnroll; ++i) asm volatile(ASM : "+v"(a[i]) : __VA_ARGS__); \
        }                                                                                             \
        const unsigned long long s1 = __builtin_readcyclecounter(), w1 = wall_clock64();              \
        float s = 0.0f;                                                                               \
        _Pragma("unroll") for (int i = 0; i < kUnroll; ++i) s += a[i];                                \
        if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = Clocks{s1 - s0, w1 - w0};                   \
        if (s == 123.456f) out[1].wall = (unsigned long long)s;                                       \
    }

KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2", "v"(m), "v"(c))
KERNEL(k_mul, "v_mul_f32 %0, %0, %1", "v"(m))
KERNEL(k_add, "v_add_f32 %0, %0, %1", "v"(m))
KERNEL(k_sub, "v_sub_f32 %0, %0, %1", "v"(m))
KERNEL(k_max, "v_max_f32 %0, %0, %1", "v"(m))
KERNEL(k_min, "v_min_f32 %0, %0, %1", "v"(m))
KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %2", "v"(m), "v"(c))
KERNEL(k_min3, "v_min3_f32 %0, %0, %1, %2", "v"(m), "v"(c))
KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2", "v"(m), "v"(c))
KERNEL(k_mov, "v_mov_b32 %0, %1", "v"(m))
KERNEL(k_and, "v_and_b32 %0, %0, %1", "v"(m))
KERNEL(k_xor, "v_xor_b32 %0, %0, %1", "v"(m))
KERNEL(k_shl, "v_lshlrev_b32 %0, 3, %0", "v"(m))
KERNEL(k_addu, "v_add_u32 %0, %0, %1", "v"(m))
KERNEL(k_cmp, "v_cmp_lt_f32 s[20:21], %0, %1", "v"(m) : "s20", "s21")
KERNEL(k_cnd, "v_cndmask_b32 %0, %0, %1, s[20:21]", "v"(m), "s"(0))
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1", "v"(m))
KERNEL(k_mulhi, "v_mul_hi_u32 %0, %0, %1", "v"(m))
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %1", "v"(m))
KERNEL(k_rcp, "v_rcp_f32 %0, %0", "v"(m))
KERNEL(k_sqrt, "v_sqrt_f32 %0, %0", "v"(m))
KERNEL(k_divfix, "v_div_fixup_f32 %0, %0, %1, %2", "v"(m), "v"(c))
KERNEL(k_cvt, "v_cvt_f32_u32 %0, %0", "v"(m))

template <class Kernel>
static int run(const char *name, Kernel kernel, int blocks_per_cu, Clocks *out) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const int grid = 256 * blocks_per_cu; // 256-thread blocks: one wave per SIMD per block
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, out, 1.0f);
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, out, 1.0f);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    Clocks c;
    CHECK(hipMemcpy(&c, out, sizeof c, hipMemcpyDeviceToHost));
    const double hz = (double)c.shader / (double)c.wall * 1e8;
    const double instr = (double)grid * 4 * kIters * kUnroll; // 4 waves per block
    const double rate = instr / (ms * 1e-3);
    printf("%-16s %d waves/SIMD: %.3e wave-instructions/s, clock %.2f GHz, %.2f cycles per instruction per SIMD\n", name,
           blocks_per_cu, rate, hz * 1e-9, 1024.0 * hz / rate);
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return 0;
}

int main() {
    Clocks *out = nullptr;
    CHECK(hipMalloc(&out, 64));
#define RUN(K) if (run(#K, K, w, out)) return 1;
    for (int w : {1, 2, 6, 8}) {
        RUN(k_fma) RUN(k_mul) RUN(k_add) RUN(k_sub) RUN(k_max) RUN(k_min) RUN(k_max3) RUN(k_min3) RUN(k_med3) RUN(k_mov)
        RUN(k_and) RUN(k_xor) RUN(k_shl) RUN(k_addu) RUN(k_cmp) RUN(k_cnd) RUN(k_mullo) RUN(k_mulhi) RUN(k_mul24) RUN(k_rcp)
        RUN(k_sqrt) RUN(k_divfix) RUN(k_cvt)
    }
    return 0;
}
