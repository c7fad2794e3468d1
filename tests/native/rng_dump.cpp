// Writes the first N draws of the kernels' random stream for path (pixel, sample) under `seed` to stdout as raw
// little-endian u32.  usage: rng_dump SEED PIXEL SAMPLE N
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../raytracer_amd/csrc/rt_math.h"

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const uint64_t seed = strtoull(argv[1], nullptr, 0);
    const uint32_t pixel = (uint32_t)strtoul(argv[2], nullptr, 0), sample = (uint32_t)strtoul(argv[3], nullptr, 0);
    const size_t n = strtoull(argv[4], nullptr, 0);
    rt::Rng rng = rt::rng_for_path(rt::mix64(seed), pixel, sample, 0u);
    std::vector<uint32_t> out(n);
    for (size_t i = 0; i < n; ++i) {
        if ((i & 3) == 0 && i + 1 < n) { // exercise the two-draw form as well
            rt::rng_next_2u32(rng, out[i], out[i + 1]);
            ++i;
        } else {
            out[i] = rt::rng_next_u32(rng);
        }
    }
    fwrite(out.data(), 4, n, stdout);
    return 0;
}
