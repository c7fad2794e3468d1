// bvh_build.h — BVH construction on the GPU (see bvh_build.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/rt_mi355x.h"
#include "rt_device.h"

namespace rt {

// Builds the tree of `n` objects with boxes `h_boxes` (host memory, object-list order) into `d_nodes` (device,
// 2n - 1 entries, pre-order) and writes the leaf order (object ID per leaf slot) to `h_order` (host, n
// entries).  Objects with IDs in [sphere_begin, sphere_end) are spheres (their leaves get kSphereBit).
// `*depth` receives the number of levels.  Synchronises `stream` before returning.
int build_bvh_device(hipStream_t stream, const BuildBox *h_boxes, uint32_t n, uint32_t sphere_begin, uint32_t sphere_end,
                     Node *d_nodes, uint32_t *h_order, uint32_t *depth, std::string &err);

// Fills the table of 128-byte records (ChainNode, rt_device.h) from the pre-order node table `d_nodes` of
// `n_nodes` = 2 * primitives - 1 entries; `d_chain` holds primitives - 1 records.  Asynchronous on `stream`.
void build_chain_nodes(hipStream_t stream, const Node *d_nodes, uint32_t n_nodes, ChainNode *d_chain);

} // namespace rt
