// Host-only check of rt::sample_block (csrc/sample_split.h), the split rt_render_node applies to a render's samples:
// for every GPU count and sample count given on the command line prints one line "n_gpus total begin: b0 c0 b1 c1 ..."
// after asserting that the blocks are contiguous, disjoint, in GPU order, cover [begin, begin + total) and differ in
// length by at most one.  tests/test_distributed.py compares the lines with distributed.sample_range.
//   g++ -std=c++17 -I raytracer_amd/csrc tests/native/sample_split_check.cpp -o sample_split_check
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "sample_split.h"

int main() {
    const uint32_t gpus[] = {1, 2, 3, 8}, totals[] = {0, 1, 7, 8, 4096}, begins[] = {0, 5, 1000};
    for (uint32_t n : gpus)
        for (uint32_t total : totals)
            for (uint32_t begin : begins) {
                std::printf("%u %u %u:", n, total, begin);
                uint32_t next = begin, shortest = UINT32_MAX, longest = 0;
                for (uint32_t i = 0; i < n; ++i) {
                    uint32_t b = 0, c = 0;
                    rt::sample_block(begin, total, n, i, &b, &c);
                    if (b != next) {
                        std::fprintf(stderr, "block %u of %u starts at %u, expected %u\n", i, n, b, next);
                        return 1;
                    }
                    next = b + c;
                    shortest = c < shortest ? c : shortest;
                    longest = c > longest ? c : longest;
                    std::printf(" %u %u", b, c);
                }
                std::printf("\n");
                if (next != begin + total || longest - shortest > 1u) {
                    std::fprintf(stderr, "blocks of %u samples on %u GPUs do not cover the range evenly\n", total, n);
                    return 1;
                }
            }
    return 0;
}
