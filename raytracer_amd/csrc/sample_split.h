// sample_split.h — how the samples of one render are divided among the GPUs of a node (rt_render_node).
#pragma once

#include <stdint.h>

namespace rt {

// The block of GPU `index` of `n_gpus` when samples [begin, begin + total) are split among them: contiguous blocks in
// GPU order, lengths differing by at most one, the first `total % n_gpus` blocks being the longer ones.  A GPU beyond the
// samples gets an empty block.  (raytracer_amd/distributed.py sample_range makes the same split for one process per GPU.)
inline void sample_block(uint32_t begin, uint32_t total, uint32_t n_gpus, uint32_t index, uint32_t *block_begin,
                         uint32_t *block_count) {
    const uint32_t base = total / n_gpus, extra = total % n_gpus;
    *block_begin = begin + index * base + (index < extra ? index : extra);
    *block_count = base + (index < extra ? 1u : 0u);
}

} // namespace rt
