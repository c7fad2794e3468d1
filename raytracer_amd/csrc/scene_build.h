// scene_build.h — host-side scene preparation (validation, BVH construction, flattening).
#pragma once

#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "rt_device.h"

namespace rt {

// The device tables of one scene, still in host memory.
struct HostScene {
    std::vector<Node> nodes;            // pre-order
    std::vector<Prim> prims;            // leaf order
    std::vector<TriShade> tri_shade;    // leaf order (unused entries for spheres)
    std::vector<MaterialRec> materials;
    std::vector<TextureRec> textures;
    std::vector<uint8_t> texels;
    std::vector<uint32_t> slot_of_prim; // primitive ID -> leaf slot (| kSphereBit)
    std::vector<uint32_t> prim_of_slot; // leaf slot -> primitive ID
    uint32_t depth = 0;
    uint32_t n_triangles = 0;
    uint32_t n_spheres = 0;
    bool need_uv = false;  // some material samples an image texture
    bool has_mix = false;  // some material is a Mix
};

// A tree builder that runs elsewhere (the GPU): gets the objects' boxes in object-list order and the ID range
// of the spheres, keeps the node table to itself and returns the leaf order (object ID per leaf slot) and the
// number of levels.
using ExternalTreeBuilder = std::function<int(const BuildBox *boxes, uint32_t n, uint32_t sphere_begin, uint32_t sphere_end,
                                              std::vector<uint32_t> &order, uint32_t &depth, std::string &err)>;

// Validates `desc` and fills `out`.  Returns an rt_status; on failure `err` holds the message.  With `external`
// and at least `external_min_objects` objects the tree is built by it instead of the host builder and
// `out.nodes` stays empty.
int build_host_scene(const rt_scene_desc *desc, HostScene &out, std::string &err, const ExternalTreeBuilder *external = nullptr,
                     uint64_t external_min_objects = 0);

// The walk table (WalkNode, rt_device.h) of the pre-order table `nodes` (n entries): `out` receives the n records,
// the first `*region_nodes` of them the nodes nearest the root, level by level and left to right within a level (so
// that every prefix of the region is itself a treetop), at most `max_region_nodes`; the others follow in pre-order.
// Fails (false, message in `err`) when byte offsets into the table do not fit 31 bits or a leaf slot 30 bits.
bool build_walk_table(const Node *nodes, size_t n, size_t max_region_nodes, std::vector<WalkNode> &out, size_t *region_nodes,
                      std::string &err);

} // namespace rt
