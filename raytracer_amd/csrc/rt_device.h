// rt_device.h — layout of the scene tables and wavefront state in HBM.  Shared by host and device code.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

constexpr uint32_t kInterior = 0xFFFFFFFFu; // Node::leaf of an interior node
constexpr uint32_t kSphereBit = 0x80000000u; // set in Node::leaf / hit slot for a sphere leaf
constexpr uint32_t kMissSlot = 0xFFFFFFFFu;

// One BVH node, 32 bytes = two 16-byte loads.  Nodes are stored in pre-order (the order
// BvhNode::hit visits them, src/bvh.rs:295-317): the left child of node i is node i + 1 and `skip` is the
// node that follows i's whole subtree, so "box hit -> i + 1, box miss -> skip" walks the tree in exactly the
// reference's always-left-first order without a stack.  A leaf's `skip` is i + 1.
// The two planes of an axis are neighbours so that a 16-byte load puts them in one register pair, the operand
// shape of the packed binary32 subtract and multiply of the slab test.
struct alignas(16) Node {
    float xlo, xhi, ylo, yhi; // first 16 bytes
    float zlo, zhi;
    uint32_t skip;
    uint32_t leaf; // kInterior, or index into the leaf-ordered primitive table (| kSphereBit)
    void set_box(const float lo[3], const float hi[3]) {
        xlo = lo[0], xhi = hi[0], ylo = lo[1], yhi = hi[1], zlo = lo[2], zhi = hi[2];
    }
    void get_box(float lo[3], float hi[3]) const {
        lo[0] = xlo, lo[1] = ylo, lo[2] = zlo, hi[0] = xhi, hi[1] = yhi, hi[2] = zhi;
    }
};
static_assert(sizeof(Node) == 32, "Node must be 32 bytes");

// The same tree once more for the stack traversal: one 128-byte record (one cache line: the unit every miss of
// a scattered read costs, whatever part of it is used) per INTERIOR node X, indexed by X's rank among the interior
// nodes in pre-order.  BvhNode::hit tests both children's boxes whenever it enters a node (src/bvh.rs:302-313), so
// the record holds the boxes of X's children L and R in full, and, while they can be told, the boxes of all four
// grandchildren: the children of L and the children of R.  A child box shares at least one of each pair of
// opposite planes with its parent's box (a union's extreme is attained by one of its parts), so a pair of children
// is stored as six "new" planes plus six bits saying which child inherits the parent's plane and which takes the
// new one; the boxes come out bit for bit.  Where a pair does not have that property (boxes padded differently,
// src/aabb.rs:78-93) the record does not hold it and the child is entered through its own record.  One fetch thus
// serves two levels of the tree: a walk that starts at the root reads the records of every other level only.
//   q0 = L.xlo, L.xhi, L.ylo, L.yhi     q1 = L.zlo, L.zhi, R.zlo, R.zhi     q2 = R.xlo, R.xhi, R.ylo, R.yhi
//   q3 = ref L, ref R, meta, 0          meta: bits 0-5 inherit mask of (LL, LR), bits 8-13 of (RL, RR),
//                                             bit 16: (LL, LR) held, bit 17: (RL, RR) held
//   q4 = new planes of (LL, LR): xlo, xhi, ylo, yhi       q5 = their zlo, zhi | zlo, zhi of (RL, RR)
//   q6 = new planes of (RL, RR): xlo, xhi, ylo, yhi       q7 = ref LL, ref LR, ref RL, ref RR
// Plane k of the left child of a pair is the parent's when bit k of the mask is set and the new plane otherwise;
// the right child takes the other one.  A child reference is the record index of an interior node, or
// kRefLeaf | slot in the primitive table (| kRefSphere for a sphere).
struct alignas(128) ChainNode {
    float lx[2], ly[2];
    float lz[2], rz[2];
    float rx[2], ry[2];
    uint32_t left, right, meta, pad;
    float n1xy[4];
    float n1z[2], n2z[2];
    float n2xy[4];
    uint32_t ll, lr, rl, rr;
};
static_assert(sizeof(ChainNode) == 128, "ChainNode must be 128 bytes");
constexpr uint32_t kRefLeaf = 0x80000000u;
constexpr uint32_t kRefSphere = 0x40000000u;
constexpr uint32_t kRefIndex = 0x3FFFFFFFu;
constexpr uint32_t kRefNone = 0xFFFFFFFFu;
// Set in the reference of an interior node whose record is read from the block's LDS copy of the tree's top records
// (SceneDev::chain_top): the low bits are then the record's slot in that copy.  Only references stored IN that copy carry
// it (and the root's, set by the kernel); the records in device memory never do.
constexpr uint32_t kRefTop = 0x20000000u;

// The same tree a third time, for the one-lane-per-path walk over scenes that sit in the L2s: the 32-byte nodes of the
// pre-order table with BOTH successors explicit, as byte offsets into this table (what an LDS read and a
// base-plus-offset global read want as their address):
//   skip = where BvhNode::hit goes on after this node's box is missed, or after a leaf (the pre-order table's skip)
//   leaf = interior node: where it goes on after the box is hit (the left child);
//          leaf: kRefLeaf | kRefSphere (for a sphere) | slot in the primitive table
// kRefNone as a successor ends the walk.  With explicit successors the order of the records is free: the table starts
// with the "treetop", the nodes nearest the root taken level by level (so a node's parent is in it whenever the node
// is), as many as the LDS budget allows, in pre-order among themselves; the other nodes follow in pre-order.  Every
// block stages the treetop in LDS and a lane reads node `cur` from LDS when cur < top_bytes, else through the caches.
// Visiting order, box tests and primitive tests are the pre-order walk's, hence the reference's.
typedef Node WalkNode;

// An object's own bounding box (`Hittable::bounding_box`), already padded; input of the tree builders.
struct BuildBox {
    float lo[3];
    float hi[3];
};

// One primitive of the intersection table, 48 bytes, stored in leaf (pre-order) order.
//   triangle: a = v0.xyz | prim_id, b = e1.xyz | material, c = e2.xyz | flags    (e1 = v1 - v0, e2 = v2 - v0,
//             the same subtractions TriangleRef::hit performs per call, src/object/triangle_mesh.rs:144-145)
//   sphere:   a = center.xyz | radius, b = prim_id, material, 0, 0
struct alignas(16) Prim {
    float a[4];
    float b[4];
    float c[4];
};
static_assert(sizeof(Prim) == 48, "Prim must be 48 bytes");

constexpr uint32_t kTriHasNormals = 1u; // all three vertex normals exist (else n0 = geometric normal)
constexpr uint32_t kTriHasUvs = 2u;

// Shading attributes of a triangle, 64 bytes, same order as the Prim table.
struct alignas(16) TriShade {
    float n0[3];
    float n1[3];
    float n2[3];
    float uv0[2];
    float uv1[2];
    float uv2[2];
    uint32_t flags;
};
static_assert(sizeof(TriShade) == 64, "TriShade must be 64 bytes");

// `enum Material` (src/material.rs:141-148) as a tagged record.  When the material's texture is a
// SolidColor its colour is copied into `rgb` and `solid` is 1, so the common case needs no texture fetch.
struct alignas(16) MaterialRec {
    uint32_t kind;
    int32_t texture;
    float param;
    uint32_t solid;
    float rgb[3];
    int32_t left;
    int32_t right;
    uint32_t pad[3];
};
static_assert(sizeof(MaterialRec) == 48, "MaterialRec must be 48 bytes");

// `enum Texture` (src/texture.rs:140-144).  Image texels live in one RGB8 blob at `pixel_offset`.
struct alignas(16) TextureRec {
    uint32_t kind;
    float rgb[3];
    float inv_scale;
    int32_t even;
    int32_t odd;
    uint32_t width;
    uint32_t height;
    uint32_t pad;
    uint64_t pixel_offset;
};
static_assert(sizeof(TextureRec) == 48, "TextureRec must be 48 bytes");

// Device pointers of one uploaded scene.
struct SceneDev {
    const Node *nodes;
    const Prim *prims;
    const TriShade *tri_shade;
    const MaterialRec *materials;
    const TextureRec *textures;
    const uint8_t *texels;
    const uint32_t *slot_of_prim; // primitive ID -> slot in `prims` (object-list order for World::hit)
    const ChainNode *chain;       // n_prims - 1 records; nullptr when the table was not built
    const ChainNode *chain_top;   // copies of the chain records nearest the root, level by level, child references into
                                  // the copy itself tagged with kRefTop; staged in LDS by the lanes kernel
    uint32_t n_chain_top;         // records in chain_top (0: none)
    const WalkNode *walk_nodes;   // n_nodes records, treetop first; nullptr when the table was not built
    uint32_t top_bytes;           // bytes of walk_nodes that form the treetop (a multiple of 32)
    uint32_t prims_in_lds;        // the lanes kernel also stages the primitive table in LDS, behind the treetop
    uint32_t root_ref;            // child reference of the root (a leaf reference for a one-primitive scene)
    uint32_t depth;               // levels of the tree; the chain walk's stack holds at most 2 entries per interior level
    uint32_t n_nodes;
    uint32_t n_prims;
    uint32_t n_materials;
    uint32_t n_textures;
};

// `Camera` (src/camera.rs:6-17) + the per-call constants of the integrator, passed by value to kernels.
struct RenderParams {
    float center[3];
    float pixel_00_loc[3];
    float pixel_delta_u[3];
    float pixel_delta_v[3];
    float defocus_disk_u[3];
    float defocus_disk_v[3];
    float defocus_angle;
    float z_near;
    float z_far;
    float background[3];
    uint32_t width;
    uint32_t height;
    uint32_t npix;
    uint32_t max_depth;
    uint32_t mode;
    uint32_t sample_base; // global index of the batch's first sample
    uint32_t batch_samples;
    uint32_t tiles_x; // width / 8 when camera rays are dealt in 8 x 8 pixel tiles, 0 when in row order
    uint64_t seed_mix;
};

// Wavefront state, one entry per path slot (slot = sample_in_batch * npix + pixel).
//   ray_o  = origin.xyz    | k     (index of the path's next random draw)
//   ray_d  = direction.xyz | meta  (depth in bits 0..15, sample_in_batch in bits 16..31)
//   hit    = t, slot (| kSphereBit, kMissSlot on a miss), u, v
//   beta   = throughput.xyz | unused
//   radiance = l.xyz | unused
struct PathState {
    float4 *ray_o;
    float4 *ray_d;
    float4 *hit;
    float4 *beta;
    float4 *radiance;
};

// A queue of path slots segmented by workgroup: block b owns items[b * cap .. b * cap + counts[b]).
// items == nullptr denotes the identity queue over [0, n).
struct QueueView {
    const uint32_t *items;
    const uint32_t *counts;
    uint32_t cap;
    uint32_t n;
};

// Memory of the per-lane traversal stacks beyond their LDS part: entry e of thread t (numbered over the whole
// grid) is spill[e * stride + t].
struct StackSpill {
    uint2 *spill;
    uint32_t stride;
};

struct ExtendCounters {
    unsigned long long node_visits;
    unsigned long long triangle_tests;
    unsigned long long sphere_tests;
};

} // namespace rt
