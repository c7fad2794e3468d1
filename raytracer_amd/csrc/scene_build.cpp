// scene_build.cpp — host-side scene preparation: validation, the reference's BVH construction rule and the
// flattening of everything into the device tables of rt_device.h.
#include "scene_build.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <limits>
#include <thread>

#include "rt_math.h"

namespace rt {

namespace {

// ---- Range / Aabb construction (src/range.rs:16-21,35-37,54-60; src/aabb.rs:26-93,129-137) -------------
using Box = BuildBox;

// Runs fn(begin, end) over [0, n) on the host's cores (one call when the range is small).
template <class Fn> void parallel_for(uint64_t n, Fn fn) {
    unsigned threads = std::thread::hardware_concurrency();
    if (threads > 32) threads = 32;
    if (threads < 2 || n < (1u << 16)) {
        fn((uint64_t)0, n);
        return;
    }
    std::vector<std::thread> pool;
    const uint64_t chunk = (n + threads - 1) / threads;
    for (unsigned t = 0; t < threads; ++t) {
        const uint64_t b = t * chunk, e = std::min(n, b + chunk);
        if (b < e) pool.emplace_back([=] { fn(b, e); });
    }
    for (auto &t : pool) t.join();
}

inline void pad_axis(float &lo, float &hi) {
    const float delta = 0.001f;
    if (hi - lo < delta) { // Range::size() < delta -> Range::expand(delta)
        const float padding = delta / 2.0f;
        lo = lo - padding;
        hi = hi + padding;
    }
}

inline Box padded(Box b) { // Aabb::pad_to_minimums
    for (int a = 0; a < 3; ++a) pad_axis(b.lo[a], b.hi[a]);
    return b;
}

inline Box box_from_points(V3 p, V3 q) { // Aabb::from_points
    const float pa[3] = {p.x, p.y, p.z}, qa[3] = {q.x, q.y, q.z};
    Box b;
    for (int a = 0; a < 3; ++a) {
        if (pa[a] <= qa[a]) {
            b.lo[a] = pa[a];
            b.hi[a] = qa[a];
        } else {
            b.lo[a] = qa[a];
            b.hi[a] = pa[a];
        }
    }
    return padded(b);
}

inline Box box_union(const Box &p, const Box &q) { // Aabb::from_boxes over Range::from_ranges
    Box b;
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = p.lo[a] <= q.lo[a] ? p.lo[a] : q.lo[a];
        b.hi[a] = p.hi[a] >= q.hi[a] ? p.hi[a] : q.hi[a];
    }
    return padded(b);
}

inline Box box_empty() { // Aabb::EMPTY
    const float inf = std::numeric_limits<float>::infinity();
    return Box{{inf, inf, inf}, {-inf, -inf, -inf}};
}

inline int longest_axis(const Box &b) { // Aabb::longest_axis: x is only ever compared with z
    const float sx = b.hi[0] - b.lo[0], sy = b.hi[1] - b.lo[1], sz = b.hi[2] - b.lo[2];
    if (sx > sz) return 0;
    if (sy > sz) return 1;
    return 2;
}

struct SortItem {
    float key;
    uint32_t id;
};

// Reverse(OrderedFloat(min)) (src/bvh.rs:264-266): descending, NaN greatest and equal to itself.  The
// reference's unstable parallel sort leaves the order of equal keys unspecified; ascending primitive ID is
// used here so that the tree is a pure function of the object list.
inline bool sort_before(const SortItem &p, const SortItem &q) {
    const bool pn = p.key != p.key, qn = q.key != q.key;
    if (pn != qn) return pn;
    if (!pn) {
        if (p.key > q.key) return true;
        if (p.key < q.key) return false;
    }
    return p.id < q.id;
}

struct Builder {
    const std::vector<Box> &obj_boxes;
    std::vector<Node> &nodes;
    std::vector<uint32_t> &order; // object IDs; ends up in leaf (pre-order) order
    std::atomic<uint32_t> max_level{0};
    std::atomic<int> spare_threads;

    Builder(const std::vector<Box> &b, std::vector<Node> &n, std::vector<uint32_t> &o, int threads)
        : obj_boxes(b), nodes(n), order(o), spare_threads(threads) {}

    // BvhNode::from_objects (src/bvh.rs:248-285) for order[lo, lo+len), written at pre-order index `at`.
    void build(uint32_t lo, uint32_t len, uint32_t at, uint32_t level) {
        uint32_t seen = max_level.load(std::memory_order_relaxed);
        while (level > seen && !max_level.compare_exchange_weak(seen, level, std::memory_order_relaxed)) {
        }
        Node &n = nodes[at];
        if (len == 1) {
            const Box &b = obj_boxes[order[lo]];
            n.set_box(b.lo, b.hi);
            n.skip = at + 1;
            n.leaf = lo; // slot in the leaf-ordered tables; the sphere bit is added by the caller
            return;
        }
        Box bbox = box_empty(); // Aabb::from_objects folds in the current list order (src/aabb.rs:61-68)
        for (uint32_t i = 0; i < len; ++i) bbox = box_union(bbox, obj_boxes[order[lo + i]]);
        const int axis = longest_axis(bbox);
        std::vector<SortItem> items(len);
        for (uint32_t i = 0; i < len; ++i) items[i] = SortItem{obj_boxes[order[lo + i]].lo[axis], order[lo + i]};
        std::sort(items.begin(), items.end(), sort_before);
        for (uint32_t i = 0; i < len; ++i) order[lo + i] = items[i].id;
        std::vector<SortItem>().swap(items);
        n.set_box(bbox.lo, bbox.hi);
        n.skip = at + 2 * len - 1;
        n.leaf = kInterior;
        const uint32_t mid = len / 2;
        // left subtree: pre-order slots [at+1, at+2*mid); right subtree starts at at + 2*mid
        bool spawn = false;
        if (len >= (1u << 15)) {
            if (spare_threads.fetch_sub(1) > 0)
                spawn = true;
            else
                spare_threads.fetch_add(1);
        }
        if (spawn) {
            auto fut = std::async(std::launch::async, [this, lo, mid, at, level] { build(lo, mid, at + 1, level + 1); });
            build(lo + mid, len - mid, at + 2 * mid, level + 1);
            fut.get();
            spare_threads.fetch_add(1);
        } else {
            build(lo, mid, at + 1, level + 1);
            build(lo + mid, len - mid, at + 2 * mid, level + 1);
        }
    }
};

int fail(std::string &err, int code, const char *msg) {
    err = msg;
    return code;
}

int validate_texture(const rt_scene_desc *d, int32_t ti, uint32_t depth, std::string &err) {
    if (ti < 0 || (uint32_t)ti >= d->num_textures) return fail(err, RT_ERR_INVALID_ARGUMENT, "texture index out of range");
    if (depth > d->num_textures) return fail(err, RT_ERR_INVALID_ARGUMENT, "cyclic checkerboard texture");
    const rt_texture &t = d->textures[ti];
    if (t.kind == RT_TEX_CHECKER) {
        if (int rc = validate_texture(d, t.even, depth + 1, err)) return rc;
        return validate_texture(d, t.odd, depth + 1, err);
    }
    if (t.kind > RT_TEX_IMAGE) return fail(err, RT_ERR_INVALID_ARGUMENT, "unknown texture kind");
    if (t.kind == RT_TEX_IMAGE && (uint64_t)t.width * t.height != 0 && !t.pixels)
        return fail(err, RT_ERR_INVALID_ARGUMENT, "image texture without pixels");
    return RT_OK;
}

int validate_material(const rt_scene_desc *d, int32_t mi, int depth, std::string &err) {
    if (mi < 0 || (uint32_t)mi >= d->num_materials) return fail(err, RT_ERR_INVALID_ARGUMENT, "material index out of range");
    const rt_material &m = d->materials[mi];
    switch (m.kind) {
    case RT_MAT_LAMBERTIAN:
    case RT_MAT_METAL:
    case RT_MAT_DIFFUSE_LIGHT:
        if (m.texture < 0 || (uint32_t)m.texture >= d->num_textures)
            return fail(err, RT_ERR_INVALID_ARGUMENT, "texture index out of range");
        return RT_OK;
    case RT_MAT_DIELECTRIC:
        return RT_OK;
    case RT_MAT_MIX:
        if (depth >= 4) return fail(err, RT_ERR_INVALID_ARGUMENT, "Mix nesting deeper than 4");
        if (int rc = validate_material(d, m.left, depth + 1, err)) return rc;
        return validate_material(d, m.right, depth + 1, err);
    case RT_MAT_TROWBRIDGE_REITZ:
        return fail(err, RT_ERR_UNSUPPORTED, "TrowbridgeReitz::scatter is todo!() in the reference");
    default:
        return fail(err, RT_ERR_INVALID_ARGUMENT, "unknown material kind");
    }
}

inline V3 load3(const float *p) { return V3{p[0], p[1], p[2]}; }
inline void store3(float *dst, V3 v) {
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
}

} // namespace

int build_host_scene(const rt_scene_desc *d, HostScene &out, std::string &err, const ExternalTreeBuilder *external,
                     uint64_t external_min_objects) {
    if (!d) return fail(err, RT_ERR_INVALID_ARGUMENT, "null scene description");
    if ((d->num_meshes && !d->meshes) || (d->num_spheres && !d->spheres) || (d->num_materials && !d->materials) ||
        (d->num_textures && !d->textures))
        return fail(err, RT_ERR_INVALID_ARGUMENT, "null table in scene description");
    uint64_t n_tri = 0;
    for (uint32_t i = 0; i < d->num_meshes; ++i) n_tri += d->meshes[i].num_faces;
    const uint64_t n_obj = n_tri + d->num_spheres;
    if (n_obj == 0)
        return fail(err, RT_ERR_INVALID_ARGUMENT, "empty scene (BvhNode::from_objects never terminates on it)");
    if (n_obj > 0x3FFFFFFFull) return fail(err, RT_ERR_INVALID_ARGUMENT, "too many primitives");
    for (uint32_t i = 0; i < d->num_textures; ++i)
        if (int rc = validate_texture(d, (int32_t)i, 0, err)) return rc;
    for (uint32_t i = 0; i < d->num_materials; ++i)
        if (int rc = validate_material(d, (int32_t)i, 0, err)) return rc;
    for (uint32_t i = 0; i < d->num_meshes; ++i) {
        const rt_mesh &m = d->meshes[i];
        if (m.material < 0 || (uint32_t)m.material >= d->num_materials)
            return fail(err, RT_ERR_INVALID_ARGUMENT, "mesh material out of range");
        if ((m.num_faces && !m.indices) || (m.num_vertices && !m.positions) || (m.num_normals && !m.normals) ||
            (m.num_uvs && !m.uvs))
            return fail(err, RT_ERR_INVALID_ARGUMENT, "null mesh buffer");
        for (uint64_t k = 0; k < 3ull * m.num_faces; ++k) {
            if (m.indices[k] >= m.num_vertices) return fail(err, RT_ERR_INVALID_ARGUMENT, "vertex index out of range");
            if (m.num_uvs != 0 && m.indices[k] >= m.num_uvs) return fail(err, RT_ERR_INVALID_ARGUMENT, "uv index out of range");
        }
    }
    for (uint32_t i = 0; i < d->num_spheres; ++i)
        if (d->spheres[i].material < 0 || (uint32_t)d->spheres[i].material >= d->num_materials)
            return fail(err, RT_ERR_INVALID_ARGUMENT, "sphere material out of range");

    // RT_BUILD_TIMING=1 prints how long each phase of the host build takes
    const bool timing = std::getenv("RT_BUILD_TIMING") != nullptr;
    auto t_phase = std::chrono::steady_clock::now();
    auto phase_done = [&](const char *name) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[rt build] %-28s %8.1f ms\n", name, std::chrono::duration<double, std::milli>(now - t_phase).count());
        t_phase = now;
    };
    phase_done("validation");

    // ---- materials and textures ----
    out.need_uv = false;
    out.has_mix = false;
    out.textures.resize(d->num_textures);
    out.texels.clear();
    for (uint32_t i = 0; i < d->num_textures; ++i) {
        const rt_texture &t = d->textures[i];
        TextureRec r{};
        r.kind = t.kind;
        std::memcpy(r.rgb, t.rgb, sizeof r.rgb);
        r.inv_scale = t.inv_scale;
        r.even = t.even;
        r.odd = t.odd;
        r.width = t.width;
        r.height = t.height;
        r.pixel_offset = out.texels.size();
        if (t.kind == RT_TEX_IMAGE) {
            out.need_uv = true;
            const size_t bytes = (size_t)t.width * t.height * 3;
            if (bytes) out.texels.insert(out.texels.end(), t.pixels, t.pixels + bytes);
        }
        out.textures[i] = r;
    }
    out.materials.resize(d->num_materials);
    for (uint32_t i = 0; i < d->num_materials; ++i) {
        const rt_material &m = d->materials[i];
        MaterialRec r{};
        r.kind = m.kind;
        r.texture = m.texture;
        r.param = m.param;
        r.left = m.left;
        r.right = m.right;
        if (m.kind == RT_MAT_MIX) out.has_mix = true;
        if ((m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_METAL || m.kind == RT_MAT_DIFFUSE_LIGHT) &&
            d->textures[m.texture].kind == RT_TEX_SOLID) {
            r.solid = 1;
            std::memcpy(r.rgb, d->textures[m.texture].rgb, sizeof r.rgb);
        }
        out.materials[i] = r;
    }

    // ---- the object list (src/scene.rs:238-242) and per-object boxes ----
    struct Obj {
        uint32_t mesh_or_sphere;
        uint32_t face;
        bool sphere;
    };
    std::vector<Obj> objs;
    objs.reserve(n_obj);
    auto push_spheres = [&] {
        for (uint32_t i = 0; i < d->num_spheres; ++i) objs.push_back(Obj{i, 0, true});
    };
    if (d->spheres_first) push_spheres();
    for (uint32_t mi = 0; mi < d->num_meshes; ++mi)
        for (uint32_t f = 0; f < d->meshes[mi].num_faces; ++f) objs.push_back(Obj{mi, f, false});
    if (!d->spheres_first) push_spheres();

    std::vector<Box> boxes(n_obj);
    parallel_for(n_obj, [&](uint64_t begin, uint64_t end) {
    for (uint64_t i = begin; i < end; ++i) {
        const Obj &o = objs[i];
        if (o.sphere) { // Sphere::new (src/object/sphere.rs:21-24)
            const rt_sphere &s = d->spheres[o.mesh_or_sphere];
            V3 c = load3(s.center), rv = V3{s.radius, s.radius, s.radius};
            boxes[i] = box_from_points(c - rv, c + rv);
        } else { // TriangleRef::bounding_box (src/object/triangle_mesh.rs:190-193)
            const rt_mesh &m = d->meshes[o.mesh_or_sphere];
            const uint32_t *ix = m.indices + 3 * (size_t)o.face;
            V3 v0 = load3(m.positions + 3 * (size_t)ix[0]), v1 = load3(m.positions + 3 * (size_t)ix[1]),
               v2 = load3(m.positions + 3 * (size_t)ix[2]);
            boxes[i] = box_union(box_from_points(v0, v1), box_from_points(v2, v2));
        }
    }
    });

    phase_done("object list and boxes");

    // ---- BVH ----
    std::vector<uint32_t> order(n_obj);
    if (external && n_obj >= external_min_objects) {
        out.nodes.clear();
        const uint32_t sphere_begin = d->spheres_first ? 0u : (uint32_t)n_tri;
        if (int rc = (*external)(boxes.data(), (uint32_t)n_obj, sphere_begin, sphere_begin + d->num_spheres, order, out.depth, err))
            return rc;
    } else {
        out.nodes.assign(2 * n_obj - 1, Node{});
        for (uint64_t i = 0; i < n_obj; ++i) order[i] = (uint32_t)i;
        int hw = (int)std::thread::hardware_concurrency();
        if (hw < 1) hw = 1;
        if (hw > 32) hw = 32;
        Builder builder(boxes, out.nodes, order, hw - 1);
        builder.build(0, (uint32_t)n_obj, 0, 1);
        out.depth = builder.max_level.load();
    }
    std::vector<Box>().swap(boxes);
    out.n_triangles = (uint32_t)n_tri;
    out.n_spheres = d->num_spheres;

    phase_done("BVH (median split)");

    // ---- leaf-ordered primitive tables ----
    out.prims.assign(n_obj, Prim{});
    out.tri_shade.assign(n_obj, TriShade{});
    out.slot_of_prim.assign(n_obj, 0);
    out.prim_of_slot = order;
    parallel_for(n_obj, [&](uint64_t begin, uint64_t end) {
    for (uint64_t slot = begin; slot < end; ++slot) {
        const uint32_t id = order[slot];
        const Obj &o = objs[id];
        Prim &p = out.prims[slot];
        if (o.sphere) {
            const rt_sphere &s = d->spheres[o.mesh_or_sphere];
            store3(p.a, load3(s.center));
            p.a[3] = s.radius;
            p.b[0] = u2f(id);
            p.b[1] = u2f((uint32_t)s.material);
            out.slot_of_prim[id] = (uint32_t)slot | kSphereBit;
        } else {
            const rt_mesh &m = d->meshes[o.mesh_or_sphere];
            const uint32_t *ix = m.indices + 3 * (size_t)o.face;
            V3 v0 = load3(m.positions + 3 * (size_t)ix[0]), v1 = load3(m.positions + 3 * (size_t)ix[1]),
               v2 = load3(m.positions + 3 * (size_t)ix[2]);
            V3 e1 = v1 - v0, e2 = v2 - v0;
            store3(p.a, v0);
            p.a[3] = u2f(id);
            store3(p.b, e1);
            p.b[3] = u2f((uint32_t)m.material);
            store3(p.c, e2);
            TriShade &sh = out.tri_shade[slot];
            uint32_t flags = 0;
            if (ix[0] < m.num_normals && ix[1] < m.num_normals && ix[2] < m.num_normals) {
                flags |= kTriHasNormals; // TriangleRef::normals (src/object/triangle_mesh.rs:114-124)
                std::memcpy(sh.n0, m.normals + 3 * (size_t)ix[0], 12);
                std::memcpy(sh.n1, m.normals + 3 * (size_t)ix[1], 12);
                std::memcpy(sh.n2, m.normals + 3 * (size_t)ix[2], 12);
            } else { // default_normal (src/object/triangle_mesh.rs:204-209): the same value at every hit
                store3(sh.n0, normalize(cross(e1, e2)));
            }
            if (m.num_uvs != 0) {
                flags |= kTriHasUvs;
                std::memcpy(sh.uv0, m.uvs + 2 * (size_t)ix[0], 8);
                std::memcpy(sh.uv1, m.uvs + 2 * (size_t)ix[1], 8);
                std::memcpy(sh.uv2, m.uvs + 2 * (size_t)ix[2], 8);
            }
            sh.flags = flags;
            p.c[3] = u2f(flags);
            out.slot_of_prim[id] = (uint32_t)slot;
        }
    }
    });
    // leaves: add the sphere bit now that the tables are known (an external builder sets it itself)
    for (Node &n : out.nodes)
        if (n.leaf != kInterior && objs[order[n.leaf]].sphere) n.leaf |= kSphereBit;
    phase_done("leaf-ordered tables");
    return RT_OK;
}

bool build_walk_table(const Node *nodes, size_t n, size_t max_region_nodes, std::vector<WalkNode> &out, size_t *region_nodes,
                      std::string &err) {
    if (n == 0 || n > (1ull << 26)) {
        err = "the walk table addresses its nodes by 31-bit byte offsets: at most 2^26 nodes";
        return false;
    }
    // levels: in pre-order a node's children are i + 1 and the node after the left child's subtree
    std::vector<uint32_t> level(n, 0u);
    uint32_t deepest = 0;
    for (size_t i = 0; i < n; ++i) {
        if (nodes[i].leaf != kInterior) continue;
        const size_t left = i + 1, right = left < n ? nodes[left].skip : n;
        if (right >= n) {
            err = "malformed node table";
            return false;
        }
        level[left] = level[right] = level[i] + 1u;
        deepest = std::max(deepest, level[i] + 1u);
    }
    // the region: whole levels while they fit, then the leftmost nodes of the next one
    std::vector<size_t> per_level(deepest + 1u, 0);
    for (size_t i = 0; i < n; ++i) ++per_level[level[i]];
    const size_t budget = std::min(n, max_region_nodes);
    std::vector<size_t> first_of_level(deepest + 2u, 0); // region position of a level's first node
    uint32_t partial = 0;                                // the level that is taken in part (or deepest + 1)
    size_t partial_count = 0, taken = 0;
    for (partial = 0; partial <= deepest; ++partial) {
        first_of_level[partial] = taken;
        if (taken + per_level[partial] > budget) {
            partial_count = budget - taken;
            break;
        }
        taken += per_level[partial];
    }
    const size_t region = taken + partial_count;
    // new position of every node
    std::vector<uint32_t> place(n);
    std::vector<size_t> seen(deepest + 1u, 0);
    size_t below = region;
    for (size_t i = 0; i < n; ++i) {
        const uint32_t l = level[i];
        const size_t rank = seen[l]++;
        if (l < partial || (l == partial && rank < partial_count))
            place[i] = (uint32_t)(first_of_level[l] + rank);
        else
            place[i] = (uint32_t)below++;
    }
    out.assign(n, WalkNode{});
    for (size_t i = 0; i < n; ++i) {
        WalkNode w = nodes[i];
        w.skip = nodes[i].skip < n ? place[nodes[i].skip] * 32u : kRefNone;
        if (nodes[i].leaf == kInterior) {
            w.leaf = place[i + 1] * 32u;
        } else {
            const uint32_t slot = nodes[i].leaf & ~kSphereBit;
            if (slot > kRefIndex) {
                err = "the walk table holds 30-bit leaf slots";
                return false;
            }
            w.leaf = kRefLeaf | ((nodes[i].leaf & kSphereBit) ? kRefSphere : 0u) | slot;
        }
        out[place[i]] = w;
    }
    if (region_nodes) *region_nodes = region;
    return true;
}

} // namespace rt
