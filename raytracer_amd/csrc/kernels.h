// kernels.h — host-callable launchers of the wavefront kernels (kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "rt_device.h"

namespace rt {

// Every launcher enqueues on `s` and returns immediately.  `max_blocks` caps the persistent grid; queues
// longer than the grid are covered by grid-stride loops.  `count_ptr`, when non-null, points at the queue
// length in device memory and `count_bound` is only an upper bound used to size the grid; when null,
// `count_bound` is the exact length.

void launch_generate(hipStream_t s, uint32_t max_blocks, const RenderParams &P, const PathState &st, uint32_t n_slots);

void launch_extend(hipStream_t s, uint32_t max_blocks, const SceneDev &sc, const float4 *ray_o, const float4 *ray_d,
                   float4 *hits, const uint32_t *queue, const uint32_t *count_ptr, uint32_t count_bound, float tmin,
                   float tmax, bool use_bvh, ExtendCounters *counters);

void launch_shade(hipStream_t s, uint32_t max_blocks, const SceneDev &sc, const RenderParams &P, const PathState &st,
                  const uint32_t *queue_in, const uint32_t *count_ptr, uint32_t count_bound, uint32_t *queue_out,
                  uint32_t *count_out, bool need_uv);

void launch_accumulate(hipStream_t s, uint32_t max_blocks, const RenderParams &P, const float4 *radiance, float *accum);

void launch_pack_rays(hipStream_t s, uint32_t max_blocks, const float *rays, uint32_t n, float4 *ray_o, float4 *ray_d);

void launch_finalize_hits(hipStream_t s, uint32_t max_blocks, const SceneDev &sc, const float4 *ray_o,
                          const float4 *ray_d, const float4 *hits, uint32_t n, void *out);

} // namespace rt
