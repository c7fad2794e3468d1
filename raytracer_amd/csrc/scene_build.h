// scene_build.h — host-side scene preparation (validation, BVH construction, flattening).
#pragma once

#include <stdint.h>

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

// Validates `desc` and fills `out`.  Returns an rt_status; on failure `err` holds the message.
int build_host_scene(const rt_scene_desc *desc, HostScene &out, std::string &err);

} // namespace rt
