// kernels.h — host-callable launchers of the wavefront kernels (kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "rt_device.h"

namespace rt {

// Every launcher enqueues on `s` and returns immediately.  `max_blocks` caps the grid of the kernels that run
// plain grid-stride loops.  extend and shade walk a segmented queue (QueueView, rt_device.h) and must be
// launched with the same `grid` for every bounce of a batch.

void launch_generate(hipStream_t s, uint32_t max_blocks, const RenderParams &P, const PathState &st, uint32_t n_slots);

// `lds_bytes` != 0 stages the node and primitive tables in LDS (see launch_bounce).
// Closest hits with one lane per ray (k_extend_lanes): the stack traversal over the chain records when `chain`, else
// the stackless walk; `spill` as for launch_paths_lanes.  Blocks of 1024 threads when `chain` (stride = grid * 1024), else
// of 256.  hits[i] belongs to ray i.
void launch_extend_lanes(hipStream_t s, uint32_t grid, const SceneDev &sc, const float4 *ray_o, const float4 *ray_d, float4 *hits,
                         uint32_t n, float tmin, float tmax, const StackSpill &spill, bool chain);
void launch_extend(hipStream_t s, uint32_t grid, const SceneDev &sc, const float4 *ray_o, const float4 *ray_d,
                   float4 *hits, const QueueView &q, float tmin, float tmax, bool use_bvh, ExtendCounters *counters,
                   uint32_t lds_bytes);
// Closest hits of rays [0, n) by the fused render kernels' walks over their LDS copies of the scene (48-byte sorted
// nodes, or eight copies per node when `octants`), counting box and primitive tests into `counters` when given: a
// test vehicle for those walks.  False (nothing launched) when the copy does not fit 150 KB of LDS.
bool launch_extend_plain(hipStream_t s, uint32_t grid, const SceneDev &sc, const float4 *ray_o, const float4 *ray_d, float4 *hits,
                         uint32_t n, float tmin, float tmax, ExtendCounters *counters, bool octants);

// Shades the items of `q`; block b appends its surviving slots at queue_out[b * q.cap ...], stores their
// number in counts_out[b] and in early_out[b] the number of paths it ended because their next ray misses the
// scene's root box (those rays count as traced in the next bounce).
// `has_mix` selects the variant that evaluates Mix materials.
void launch_shade(hipStream_t s, uint32_t grid, const SceneDev &sc, const RenderParams &P, const PathState &st,
                  const QueueView &q, uint32_t *queue_out, uint32_t *counts_out, uint32_t *early_out, bool need_uv,
                  bool has_mix);

// One whole path segment per item: camera ray generation (`first`), closest hit, shading, compaction.
// `lds_bytes` != 0 stages the node and primitive tables in that many bytes of LDS (2 * 16 * n_nodes +
// 3 * 16 * n_prims); the caller decides whether they fit.
void launch_bounce(hipStream_t s, uint32_t grid, const SceneDev &sc, const RenderParams &P, const PathState &st,
                   const QueueView &q, uint32_t *queue_out, uint32_t *counts_out, uint32_t *early_out, bool first,
                   bool need_uv, uint32_t lds_bytes);

// totals[r] = sum over b of counts[r * n_blocks + b] + early[r * n_blocks + b] for r in [0, n_rows).
// All bounces of a batch in one launch (k_paths); `counts` / `early` rows 1 .. max_depth must be zero.
void launch_paths(hipStream_t s, uint32_t grid, const SceneDev &sc, const RenderParams &P, const PathState &st, uint32_t n_slots,
                  uint32_t cap, uint32_t *queue0, uint32_t *queue1, uint32_t *counts, uint32_t *early, bool need_uv,
                  bool has_mix, uint32_t lds_bytes, bool big_blocks, bool eight_waves);
// Entries per lane of the stack traversal's stack that live in LDS; deeper ones go to the spill area
// (10 M soup: 4 entries 0.56, 8 0.62, 16 0.63 Grays/s; 32 KB of LDS per 256-thread block at 16).
#ifndef RT_STACK_K
#define RT_STACK_K 16
#endif
constexpr int kStackEntriesInLds = RT_STACK_K;
// One lane per path (k_paths_lanes): the whole batch in one launch without queues or path state in device
// memory, for scenes read from HBM; needs sc.chain.  `ray_counts[b]` receives the rays block b traced beyond each
// path's first.  `waves`: waves per SIMD the register budget is set for (4, 6 or 8).
// `camera_rays_in_lockstep`: bounce 0 is traced by the per-bounce kernel first, 64 neighbouring pixels per wave
// (survivors pass through `queue`, one segment of `cap` entries per block, and the ray_o / ray_d / beta arrays of
// `st`), then one lane per surviving path; otherwise lanes take paths from the camera ray on and `queue` and those
// arrays are not touched.  `counts` / `early` are the [max_depth + 1][grid] tables, rows 1.. zeroed by the caller.
void launch_paths_lanes(hipStream_t s, uint32_t grid, const SceneDev &sc, const RenderParams &P, const PathState &st,
                        uint32_t n_slots, uint32_t cap, uint32_t *queue, uint32_t *counts, uint32_t *early, bool need_uv, bool has_mix,
                        int waves, const StackSpill &spill, bool camera_rays_in_lockstep, bool chain, uint32_t walk_threads,
                        uint32_t walk_grid);
// `chain`: traverse the chain records with the per-lane stack (needs sc.chain); otherwise the stackless walk over the
// walk table (needs sc.walk_nodes), whose treetop (sc.top_bytes) every block stages in LDS: then the lanes kernel runs
// `walk_grid` blocks of `walk_threads` threads (256, 512 or 1024), `walk_grid` dividing `grid` with a quotient of at
// most 32, while bounce 0 (when in lockstep) and the tables keep `grid` blocks of 256.
void launch_sum_counts(hipStream_t s, const uint32_t *counts, const uint32_t *early, uint32_t n_blocks, uint32_t n_rows,
                       unsigned long long *totals);

void launch_accumulate(hipStream_t s, uint32_t max_blocks, const RenderParams &P, const float4 *radiance, float *accum);

void launch_pack_rays(hipStream_t s, uint32_t max_blocks, const float *rays, uint32_t n, float4 *ray_o, float4 *ray_d);

void launch_finalize_hits(hipStream_t s, uint32_t max_blocks, const SceneDev &sc, const float4 *ray_o,
                          const float4 *ray_d, const float4 *hits, uint32_t n, void *out);

// Device scratch of the allocation-free entry point (rt_trace_closest_noalloc): a file-scope array in device memory,
// kNoAllocMaxRays slots of ray origin, ray direction and hit record (48 bytes each).  Every GPU holds its own copy
// at its own address; this returns the address of the CURRENT device's copy (resolved once per device) or nullptr
// with `error` set to the HIP error string.
constexpr uint32_t kNoAllocMaxRays = 1u << 16;
float4 *noalloc_scratch_of_current_device(const char **error);

} // namespace rt
