// rt_math.h — binary32 vector arithmetic, sin/cos and the per-path random stream, shared by the HIP kernels
// and the host-side scene preparation.
//
// Everything here must round exactly like the reference's glam::Vec3A code on baseline x86-64
// (src/vec3.rs:1-3): one IEEE operation per source operation, no FMA contraction (the build passes
// -ffp-contract=off), correctly rounded division and square root.
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rt {

struct V3 {
    float x, y, z;
};

RT_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
RT_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_HD V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_HD V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
RT_HD V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
RT_HD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
// glam 0.28 dot: (x*x' + y*y') + z*z'
RT_HD float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// glam 0.28 cross: each lane is one product minus another
RT_HD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
RT_HD float length_squared(V3 a) { return dot(a, a); }
// glam 0.28 normalize: v * (1 / sqrt(v.v))
RT_HD V3 normalize(V3 a) { return a * (1.0f / sqrtf(dot(a, a))); }

// src/vec3.rs:12-14
RT_HD V3 reflect(V3 v, V3 n) { return v - (n * dot(v, n)) * 2.0f; }

// src/vec3.rs:16-21
RT_HD V3 refract(V3 uv, V3 n, float etai_over_etat) {
    float cos_theta = fminf(dot(-uv, n), 1.0f);
    V3 r_out_perp = (uv + n * cos_theta) * etai_over_etat;
    V3 r_out_parallel = n * -sqrtf(fabsf(1.0f - length_squared(r_out_perp)));
    return r_out_perp + r_out_parallel;
}

RT_HD uint32_t f2u(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
RT_HD float u2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// ------------------------------------------------------------------------------------------------------
// sin / cos for the disk sampler (src/sample.rs:22).  The reference calls the platform libm through
// f32::cos / f32::sin; on Linux that is glibc's sinf/cosf (double-precision minimax polynomials with a
// quadrant reduction at |x| >= 0.75).  This is a restatement of that published algorithm and its
// coefficient table; an exhaustive host-side comparison over every binary32 in [-1, 1] (the only range the
// sampler produces) matches glibc 2.35 bit for bit, with and without fused multiply-add.  Arguments outside
// [-1, 1] are not supported.
// Provenance: algorithm and coefficients of sysdeps/ieee754/flt-32/{s_sinf.c, s_cosf.c, s_sincosf.h,
// s_sincosf_data.c} of the GNU C Library (contributed by Arm Ltd., "Optimized Routines"; Copyright (C) 2018-2022 Free
// Software Foundation, Inc.; GNU Lesser General Public License v2.1 or later; the same routines are published by Arm
// under the MIT licence).  Written out again here, not copied; the numeric constants are theirs.
// ------------------------------------------------------------------------------------------------------
namespace sincos_detail {
constexpr double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5,
                 C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
constexpr double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
constexpr double HPI_INV = 0x1.45F306DC9C883p+23; // 2/pi * 2^24
constexpr double HPI = 0x1.921FB54442D18p0;       // pi/2

RT_HD uint32_t abstop12(float x) { return (f2u(x) >> 20) & 0x7ffu; }

// n even: sine polynomial; n odd: cosine polynomial with its coefficients multiplied by cs (+1 / -1).
RT_HD float poly(double x, double x2, double cs, int n) {
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = S2 + x2 * S3;
        double x7 = x3 * x2;
        double s = x + x3 * S1;
        return (float)(s + x7 * s1);
    }
    double x4 = x2 * x2;
    double c2 = cs * C3 + x2 * (cs * C4);
    double c1 = cs * C0 + x2 * (cs * C1);
    double x6 = x4 * x2;
    double c = c1 + x4 * (cs * C2);
    return (float)(c + x6 * c2);
}

// Quadrant reduction for 0.75 <= |y| <= 1: n is -1, 0 or 1.
RT_HD double reduce(double x, int *np) {
    double r = x * HPI_INV;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return x - n * HPI;
}
RT_HD double quadrant_sign(int n) { return ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0; }
} // namespace sincos_detail

RT_HD float sin_unit(float y) {
    using namespace sincos_detail;
    double x = y;
    if (abstop12(y) < 0x3f4u) { // |y| < 0.75
        if (abstop12(y) < 0x398u) return y; // |y| < 2^-12
        return poly(x, x * x, 1.0, 0);
    }
    int n;
    x = reduce(x, &n);
    return poly(x * quadrant_sign(n), x * x, (n & 2) ? -1.0 : 1.0, n);
}

RT_HD float cos_unit(float y) {
    using namespace sincos_detail;
    double x = y;
    if (abstop12(y) < 0x3f4u) {
        if (abstop12(y) < 0x398u) return 1.0f;
        return poly(x, x * x, 1.0, 1);
    }
    int n;
    x = reduce(x, &n);
    return poly(x * quadrant_sign(n), x * x, (n & 2) ? -1.0 : 1.0, n ^ 1);
}

// ------------------------------------------------------------------------------------------------------
// acos / atan2 for Sphere::get_uv (src/object/sphere.rs:42-52).  The reference calls the platform libm through
// f32::acos / f32::atan2; on Linux that is glibc's acosf / atan2f, which up to glibc 2.40 are the binary32
// fdlibm routines (rational approximation of asin on three ranges; atan by argument reduction to one of four
// break points and an 11-term polynomial).  This is a restatement of that published algorithm and its
// coefficient tables in binary32 arithmetic, one IEEE operation per source operation; a host-side comparison
// with glibc 2.35 (every binary32 for acosf and atanf, 10^9 argument pairs for atan2f) matches bit for bit.
// Provenance: algorithm and coefficients of sysdeps/ieee754/flt-32/{e_acosf.c, s_atanf.c, e_atan2f.c} of the GNU C
// Library, which are the FreeBSD / fdlibm binary32 routines ("Copyright (C) 1993 by Sun Microsystems, Inc.  All rights
// reserved.  Developed at SunPro, a Sun Microsystems, Inc. business.  Permission to use, copy, modify, and distribute this
// software is freely granted, provided that this notice is preserved.").  Written out again here; the constants are theirs.
// ------------------------------------------------------------------------------------------------------
namespace invtrig_detail {
RT_HD float asin_ratio(float z) {
    const float pS0 = u2f(0x3e2aaaabu), pS1 = u2f(0xbea6b090u), pS2 = u2f(0x3e4e0aa8u), pS3 = u2f(0xbd241146u),
                pS4 = u2f(0x3a4f7f04u), pS5 = u2f(0x3811ef08u);
    const float qS1 = u2f(0xc019d139u), qS2 = u2f(0x4001572du), qS3 = u2f(0xbf303361u), qS4 = u2f(0x3d9dc62eu);
    float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    return p / q;
}
} // namespace invtrig_detail

RT_HD float acos_f32(float x) {
    using namespace invtrig_detail;
    const float pi = u2f(0x40490fdau), pio2_hi = u2f(0x3fc90fdau), pio2_lo = u2f(0x33a22168u);
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) { // |x| < 0.5
        if (ix <= 0x32800000) return pio2_hi + pio2_lo;
        float r = asin_ratio(x * x);
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (hx < 0) { // x < -0.5
        float z = (1.0f + x) * 0.5f;
        float r = asin_ratio(z);
        float s = sqrtf(z);
        float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    float z = (1.0f - x) * 0.5f; // x > 0.5
    float s = sqrtf(z);
    float df = u2f(f2u(s) & 0xfffff000u);
    float c = (z - df * df) / (s + df);
    float r = asin_ratio(z);
    float w = r * s + c;
    return 2.0f * (df + w);
}

RT_HD float atan_f32(float x) {
    const float atanhi[4] = {u2f(0x3eed6338u), u2f(0x3f490fdau), u2f(0x3f7b985eu), u2f(0x3fc90fdau)};
    const float atanlo[4] = {u2f(0x31ac3769u), u2f(0x33222168u), u2f(0x33140fb4u), u2f(0x33a22168u)};
    const float aT[11] = {u2f(0x3eaaaaabu), u2f(0xbe4ccccdu), u2f(0x3e124925u), u2f(0xbde38e38u), u2f(0x3dba2e6eu),
                          u2f(0xbd9d8795u), u2f(0x3d886b35u), u2f(0xbd6ef16bu), u2f(0x3d4bda59u), u2f(0xbd15a221u),
                          u2f(0x3c8569d7u)};
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) { // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) { // |x| < 0.4375
        if (ix < 0x31000000) return x;
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) {
                id = 0;
                x = (2.0f * x - 1.0f) / (2.0f + x);
            } else {
                id = 1;
                x = (x - 1.0f) / (x + 1.0f);
            }
        } else if (ix < 0x401c0000) {
            id = 2;
            x = (x - 1.5f) / (1.0f + 1.5f * x);
        } else {
            id = 3;
            x = -1.0f / x;
        }
    }
    float z = x * x;
    float w = z * z;
    float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return hx < 0 ? -z : z;
}

RT_HD float atan2_f32(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = u2f(0x3f490fdbu), pi_o_2 = u2f(0x3fc90fdbu), pi = u2f(0x40490fdbu),
                pi_lo = u2f(0xb3bbbd2eu);
    const int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return atan_f32(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        if (m < 2) return y;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            default: return -3.0f * pi_o_4 - tiny;
            }
        }
        switch (m) {
        case 0: return 0.0f;
        case 1: return -0.0f;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60)
        z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60)
        z = 0.0f;
    else
        z = atan_f32(fabsf(y / x));
    switch (m) {
    case 0: return z;
    case 1: return u2f(f2u(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

// ------------------------------------------------------------------------------------------------------
// Per-path random stream.  The reference draws from a thread-local SmallRng seeded from OS entropy
// (src/random.rs:8-14), which cannot be reproduced; here draw k of path (pixel, sample) is a pure function
// of (seed, pixel, sample, k): 64-bit block b = mix64((sample<<42 | pixel<<16 | b) * golden + mix64(seed))
// holds draws 2b (low half) and 2b+1 (high half).  mix64 is the splitmix64 finaliser, a bijection, so
// distinct (pixel, sample, b) with b < 65536 never share a block; later blocks (see rng_block) are distinct hash
// inputs with overwhelming probability only.  The kernels carry k in 28 bits: a path's stream is defined for its
// first 2^28 draws (max_depth is at most 65535, so that is 4096 draws per bounce).
// ------------------------------------------------------------------------------------------------------
RT_HD uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct Rng {
    uint64_t seed_mix; // mix64(seed)
    uint64_t path_key; // (sample << 42) | (pixel << 16)
    uint32_t k;        // index of the next 32-bit draw
};

RT_HD Rng rng_for_path(uint64_t seed_mix, uint32_t pixel, uint32_t sample, uint32_t k) {
    Rng r;
    r.seed_mix = seed_mix;
    r.path_key = ((uint64_t)sample << 42) | ((uint64_t)(pixel & 0x3FFFFFFu) << 16);
    r.k = k;
    return r;
}

RT_HD uint64_t rng_block(const Rng &r, uint32_t b) {
    uint64_t c = r.path_key | (uint64_t)(b & 0xFFFFu);
    // Blocks beyond the 16 bits the key has room for (a path's draws from number 131072 on) are told apart by a
    // second odd multiple added to the hash input; the first 65536 blocks have b >> 16 == 0.
    return mix64(c * 0x9E3779B97F4A7C15ull + r.seed_mix + (uint64_t)(b >> 16) * 0xD6E8FEB86659FD93ull);
}

RT_HD uint32_t rng_next_u32(Rng &r) {
    uint32_t k = r.k++;
    uint64_t bits = rng_block(r, k >> 1);
    return (k & 1u) ? (uint32_t)(bits >> 32) : (uint32_t)bits;
}

// Two consecutive draws; one block evaluation when k is even (the common case).
RT_HD void rng_next_2u32(Rng &r, uint32_t &a, uint32_t &b) {
    if ((r.k & 1u) == 0) {
        uint64_t bits = rng_block(r, r.k >> 1);
        r.k += 2;
        a = (uint32_t)bits;
        b = (uint32_t)(bits >> 32);
    } else {
        a = rng_next_u32(r);
        b = rng_next_u32(r);
    }
}

// rand 0.8.5 `gen::<f32>()` (src/random.rs:12-14): 24 random bits scaled by 2^-24.
RT_HD float u32_to_unit(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }
RT_HD float rng_random(Rng &r) { return u32_to_unit(rng_next_u32(r)); }

// rand 0.8.5 `gen_range(lo..hi)` for f32 (src/random.rs:16-18): a [0,1) value with 23 random mantissa
// bits, times (hi - lo), plus lo; redrawn while the result is not below hi.
RT_HD float rng_random_range(Rng &r, float lo, float hi) {
    float scale = hi - lo;
    for (;;) {
        float value1_2 = u2f((rng_next_u32(r) >> 9) | 0x3F800000u);
        float value0_1 = value1_2 - 1.0f;
        float res = value0_1 * scale + lo;
        if (res < hi) return res;
    }
}

} // namespace rt
