// Compares rt::acos_f32 / rt::atan_f32 / rt::atan2_f32 (the functions Sphere::get_uv uses on the GPU) with the host
// libm.  usage: invtrig_check STRIDE PAIRS -> prints "<acos checked> <acos bad> <atan checked> <atan bad> <atan2 checked> <atan2 bad>"
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../raytracer_amd/csrc/rt_math.h"

static bool same(float a, float b) { return rt::f2u(a) == rt::f2u(b) || (a != a && b != b); }

int main(int argc, char **argv) {
    const uint64_t stride = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const uint64_t pairs = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1000000;
    uint64_t n1 = 0, b1 = 0, n2 = 0, b2 = 0, n3 = 0, b3 = 0;
    for (uint64_t u = 0; u <= 0xFFFFFFFFull; u += stride) {
        float x = rt::u2f((uint32_t)u);
        b1 += !same(rt::acos_f32(x), acosf(x));
        b2 += !same(rt::atan_f32(x), atanf(x));
        ++n1, ++n2;
    }
    // atan2: random bit patterns, random points of the unit sphere's equator band (what get_uv feeds it), specials
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (uint64_t i = 0; i < pairs; ++i) {
        uint64_t r = next();
        float y = rt::u2f((uint32_t)r), x = rt::u2f((uint32_t)(r >> 32));
        b3 += !same(rt::atan2_f32(y, x), atan2f(y, x));
        float yy = (float)((double)(next() >> 11) / 9007199254740992.0 * 2.0 - 1.0);
        float xx = (float)((double)(next() >> 11) / 9007199254740992.0 * 2.0 - 1.0);
        b3 += !same(rt::atan2_f32(yy, xx), atan2f(yy, xx));
        n3 += 2;
    }
    const float sp[] = {0.0f, -0.0f, 1.0f, -1.0f, INFINITY, -INFINITY, NAN, 1e-30f, -1e-30f, 1e30f, -1e30f, 0.5f, 2.0f, 1e-45f};
    for (float y : sp)
        for (float x : sp) {
            b3 += !same(rt::atan2_f32(y, x), atan2f(y, x));
            ++n3;
        }
    printf("%llu %llu %llu %llu %llu %llu\n", (unsigned long long)n1, (unsigned long long)b1, (unsigned long long)n2,
           (unsigned long long)b2, (unsigned long long)n3, (unsigned long long)b3);
    return 0;
}
