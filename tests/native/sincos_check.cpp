// Compares rt::sin_unit / rt::cos_unit (the functions the HIP shade kernel uses) with the host libm over
// binary32 values in [-1, 1].  usage: sincos_check STRIDE  -> prints "<checked> <sin mismatches> <cos mismatches>"
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../raytracer_amd/csrc/rt_math.h"

int main(int argc, char **argv) {
    const uint64_t stride = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    uint64_t checked = 0, bad_s = 0, bad_c = 0;
    for (uint64_t u = 0; u <= 0x3F800000ull; u += stride) {
        float f = rt::u2f((uint32_t)u);
        for (int sg = 0; sg < 2; ++sg) {
            float x = sg ? -f : f;
            bad_s += rt::f2u(rt::sin_unit(x)) != rt::f2u(sinf(x));
            bad_c += rt::f2u(rt::cos_unit(x)) != rt::f2u(cosf(x));
            ++checked;
        }
    }
    // the values around the quadrant switch (|x| = 0.75) and the tiny-argument cut (2^-12), exhaustively
    const uint32_t centres[2] = {0x3F400000u, 0x39800000u};
    for (uint32_t c : centres)
        for (uint32_t u = c - 4096; u <= c + 4096; ++u) {
            float x = rt::u2f(u);
            bad_s += rt::f2u(rt::sin_unit(x)) != rt::f2u(sinf(x));
            bad_c += rt::f2u(rt::cos_unit(-x)) != rt::f2u(cosf(-x));
            ++checked;
        }
    printf("%llu %llu %llu\n", (unsigned long long)checked, (unsigned long long)bad_s, (unsigned long long)bad_c);
    return 0;
}
