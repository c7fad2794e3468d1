// bvh_build.hip — the reference's BVH construction (BvhNode::from_objects, src/bvh.rs:248-285) on the GPU.
//
// The rule is top-down: a node's box is the fold of its objects' boxes in their current order
// (Aabb::from_objects, src/aabb.rs:61-68, with the 0.001 padding after every union), the split axis comes from
// that box (Aabb::longest_axis, src/aabb.rs:129-137), the objects are sorted descending by the minimum of
// their own box on that axis (equal keys in ascending primitive ID, the tie rule of the host builder) and the
// first len/2 go left.  Median splits make every level a partition of the object list into contiguous
// segments of nearly equal length, so the whole tree is built level by level:
//
//   per level:  node boxes + axis (one block per segment while segments are long, one thread when short)
//               -> 64-bit keys (descending float key | primitive ID)
//               -> radix sort of all objects by that key, then a stable radix sort by segment start, which
//                  puts every segment back in place, now internally ordered
//               -> every position computes the segment (start, length, pre-order index) of the child it
//                  now belongs to.
//
// Nodes are written straight into the pre-order table (left child at + 1, right child at + 2 * mid) and the
// object order after the last level is the leaf order.  The result is bit-identical to the host builder's.
//
// Node boxes.  Padding only ever fires while the running box is thinner than 0.001 on some axis, and the
// running box only grows, so the fold equals: fold the first few objects one by one until no axis is thin,
// then take the plain union with everything else.  Long segments use that (a parallel min/max reduction plus
// a short sequential prefix by one thread); short segments simply fold sequentially.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <string>

#include "bvh_build.h"

namespace rt {

namespace {

constexpr int kThreads = 256;
constexpr uint32_t kLongSegment = 2048; // segments at least this long get a block each

struct DBox {
    float lo[3];
    float hi[3];
};

__device__ __forceinline__ void pad_axis(float &lo, float &hi) { // Aabb::pad_to_minimums, one axis
    const float delta = 0.001f;
    if (hi - lo < delta) {
        const float padding = delta / 2.0f;
        lo = lo - padding;
        hi = hi + padding;
    }
}

__device__ __forceinline__ void union_plain(DBox &acc, const DBox &b) { // Range::from_ranges per axis
    for (int a = 0; a < 3; ++a) {
        acc.lo[a] = acc.lo[a] <= b.lo[a] ? acc.lo[a] : b.lo[a];
        acc.hi[a] = acc.hi[a] >= b.hi[a] ? acc.hi[a] : b.hi[a];
    }
}

__device__ __forceinline__ void union_padded(DBox &acc, const DBox &b) { // Aabb::from_boxes
    union_plain(acc, b);
    for (int a = 0; a < 3; ++a) pad_axis(acc.lo[a], acc.hi[a]);
}

__device__ __forceinline__ DBox empty_box() { // Aabb::EMPTY
    const float inf = __builtin_inff();
    return DBox{{inf, inf, inf}, {-inf, -inf, -inf}};
}

__device__ __forceinline__ bool thin(const DBox &b) {
    return (b.hi[0] - b.lo[0] < 0.001f) || (b.hi[1] - b.lo[1] < 0.001f) || (b.hi[2] - b.lo[2] < 0.001f);
}

__device__ __forceinline__ int longest_axis(const DBox &b) { // x is only ever compared with z
    const float sx = b.hi[0] - b.lo[0], sy = b.hi[1] - b.lo[1], sz = b.hi[2] - b.lo[2];
    if (sx > sz) return 0;
    if (sy > sz) return 1;
    return 2;
}

__device__ __forceinline__ void write_node(Node *nodes, uint32_t at, const DBox &b, uint32_t skip, uint32_t leaf) {
    float4 *out = reinterpret_cast<float4 *>(nodes + at);
    out[0] = make_float4(b.lo[0], b.hi[0], b.lo[1], b.hi[1]);
    out[1] = make_float4(b.lo[2], b.hi[2], __uint_as_float(skip), __uint_as_float(leaf));
}

// Ascending unsigned order of this value == the host comparator's order (descending key, NaN first, -0 == +0).
__device__ __forceinline__ uint32_t descending_key(float f) {
    if (f != f) return 0u;
    if (f == 0.0f) f = 0.0f;
    uint32_t u = __float_as_uint(f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~u;
}

struct Level {
    const DBox *boxes;
    uint32_t *order;   // object at each position
    uint32_t *seg_lo;  // per position: first position of its segment (its own position once it is a finished leaf)
    uint32_t *seg_len; // per position: length of its segment, 0 once it is a finished leaf
    uint32_t *seg_at;  // per position: pre-order index of its segment's node
    uint8_t *axis;     // per segment start: split axis
    Node *nodes;
    uint32_t n;
    uint32_t sphere_begin, sphere_end;
};

__global__ __launch_bounds__(kThreads) void k_init(Level lv) {
    const uint32_t p = blockIdx.x * kThreads + threadIdx.x;
    if (p >= lv.n) return;
    lv.order[p] = p;
    lv.seg_lo[p] = 0u;
    lv.seg_len[p] = lv.n;
    lv.seg_at[p] = 0u;
}

// Short segments: the thread at a segment's first position folds the segment sequentially, exactly as
// Aabb::from_objects does, or writes the leaf of a single object.
__global__ __launch_bounds__(kThreads) void k_nodes_short(Level lv) {
    const uint32_t p = blockIdx.x * kThreads + threadIdx.x;
    if (p >= lv.n) return;
    const uint32_t len = lv.seg_len[p];
    if (len == 0u || lv.seg_lo[p] != p) return;
    const uint32_t at = lv.seg_at[p];
    if (len == 1u) {
        const uint32_t id = lv.order[p];
        const uint32_t sphere = (id >= lv.sphere_begin && id < lv.sphere_end) ? kSphereBit : 0u;
        write_node(lv.nodes, at, lv.boxes[id], at + 1u, p | sphere);
        lv.seg_len[p] = 0u; // finished; seg_lo[p] == p already
        return;
    }
    DBox acc = empty_box();
    for (uint32_t k = 0; k < len; ++k) union_padded(acc, lv.boxes[lv.order[p + k]]);
    lv.axis[p] = (uint8_t)longest_axis(acc);
    write_node(lv.nodes, at, acc, at + 2u * len - 1u, kInterior);
}

// Long segments: collect the segment starts, then one block per segment.
__global__ __launch_bounds__(kThreads) void k_collect_heads(Level lv, uint32_t *heads, uint32_t *n_heads) {
    const uint32_t p = blockIdx.x * kThreads + threadIdx.x;
    if (p >= lv.n) return;
    if (lv.seg_len[p] != 0u && lv.seg_lo[p] == p) heads[atomicAdd(n_heads, 1u)] = p;
}

__global__ __launch_bounds__(kThreads) void k_nodes_long(Level lv, const uint32_t *heads, const uint32_t *n_heads) {
    if (blockIdx.x >= *n_heads) return;
    const uint32_t p = heads[blockIdx.x];
    const uint32_t len = lv.seg_len[p];
    __shared__ float s_red[kThreads / 64][6];
    __shared__ DBox s_prefix;
    DBox acc = empty_box();
    for (uint32_t k = threadIdx.x; k < len; k += kThreads) union_plain(acc, lv.boxes[lv.order[p + k]]);
    // min / max over the block: wave shuffles, then thread 0 over the per-wave results
    float v[6] = {acc.lo[0], acc.lo[1], acc.lo[2], acc.hi[0], acc.hi[1], acc.hi[2]};
    for (int sh = 32; sh > 0; sh >>= 1)
        for (int c = 0; c < 6; ++c) {
            const float o = __shfl_down(v[c], sh);
            v[c] = c < 3 ? (v[c] <= o ? v[c] : o) : (v[c] >= o ? v[c] : o);
        }
    if ((threadIdx.x & 63) == 0)
        for (int c = 0; c < 6; ++c) s_red[threadIdx.x >> 6][c] = v[c];
    if (threadIdx.x == kThreads - 1) { // the sequential prefix: fold until no axis is thin
        DBox pre = empty_box();
        for (uint32_t k = 0; k < len; ++k) {
            union_padded(pre, lv.boxes[lv.order[p + k]]);
            if (!thin(pre)) break;
        }
        s_prefix = pre;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        DBox all = s_prefix;
        for (int w = 0; w < kThreads / 64; ++w) {
            DBox part{{s_red[w][0], s_red[w][1], s_red[w][2]}, {s_red[w][3], s_red[w][4], s_red[w][5]}};
            union_plain(all, part);
        }
        lv.axis[p] = (uint8_t)longest_axis(all);
        const uint32_t at = lv.seg_at[p];
        write_node(lv.nodes, at, all, at + 2u * len - 1u, kInterior);
    }
}

// Sort keys of one level.  `stay` is the key of the second sort: the segment start (finished leaves keep
// their own position).
__global__ __launch_bounds__(kThreads) void k_keys(Level lv, unsigned long long *keys, uint32_t *stay) {
    const uint32_t p = blockIdx.x * kThreads + threadIdx.x;
    if (p >= lv.n) return;
    const uint32_t id = lv.order[p];
    const uint32_t len = lv.seg_len[p];
    uint32_t hi = 0u;
    if (len > 1u) hi = descending_key(lv.boxes[id].lo[lv.axis[lv.seg_lo[p]]]);
    keys[p] = ((unsigned long long)hi << 32) | id;
    stay[p] = lv.seg_lo[p];
}

__global__ __launch_bounds__(kThreads) void k_low_words(const unsigned long long *keys, uint32_t *ids, uint32_t n) {
    const uint32_t p = blockIdx.x * kThreads + threadIdx.x;
    if (p < n) ids[p] = (uint32_t)keys[p];
}

// After the sort: the new order, and every position moves on to the child segment it now lies in.
__global__ __launch_bounds__(kThreads) void k_split(Level lv, const uint32_t *sorted_ids) {
    const uint32_t p = blockIdx.x * kThreads + threadIdx.x;
    if (p >= lv.n) return;
    lv.order[p] = sorted_ids[p];
    const uint32_t len = lv.seg_len[p];
    if (len <= 1u) return;
    const uint32_t lo = lv.seg_lo[p], at = lv.seg_at[p], mid = len / 2u;
    if (p < lo + mid) {
        lv.seg_len[p] = mid;
        lv.seg_at[p] = at + 1u;
    } else {
        lv.seg_lo[p] = lo + mid;
        lv.seg_len[p] = len - mid;
        lv.seg_at[p] = at + 2u * mid;
    }
}

struct Scratch {
    void *ptrs[16] = {};
    int n = 0;
    ~Scratch() {
        for (int i = 0; i < n; ++i)
            if (ptrs[i]) (void)hipFree(ptrs[i]);
    }
    template <class T> bool alloc(T **out, size_t count) {
        void *p = nullptr;
        if (hipMalloc(&p, (count ? count : 1) * sizeof(T)) != hipSuccess) return false;
        ptrs[n++] = p;
        *out = static_cast<T *>(p);
        return true;
    }
};

#define BUILD_TRY(expr)                                                               \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            return e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_DEVICE;  \
        }                                                                             \
    } while (0)

} // namespace

namespace {

// Leaf slots are numbered in pre-order, so the leaves that precede node i in pre-order number slot(leftmost leaf
// under i), and the interior nodes that precede it i minus that: the record index of an interior node.
__device__ __forceinline__ uint32_t leaves_before(const Node *__restrict__ nodes, uint32_t i) {
    uint32_t leaf = nodes[i].leaf;
    while (leaf == kInterior) leaf = nodes[++i].leaf;
    return leaf & ~kSphereBit;
}

__device__ __forceinline__ uint32_t child_ref(const Node *__restrict__ nodes, uint32_t c) {
    const uint32_t leaf = nodes[c].leaf;
    if (leaf == kInterior) return c - leaves_before(nodes, c);
    return kRefLeaf | ((leaf & kSphereBit) ? kRefSphere : 0u) | (leaf & ~kSphereBit);
}

// The six planes of a node's box in the order xlo, xhi, ylo, yhi, zlo, zhi.
__device__ __forceinline__ void planes_of(const Node &n, float p[6]) {
    p[0] = n.xlo, p[1] = n.xhi, p[2] = n.ylo, p[3] = n.yhi, p[4] = n.zlo, p[5] = n.zhi;
}

// Children `a` (left) and `b` (right) of a node with planes `parent`: true when every plane of the parent is, bit
// for bit, the plane of at least one child; then `fresh[k]` is the other child's plane and bit k of `mask` says
// that the left child is the one that inherits.
__device__ __forceinline__ bool encode_pair(const float parent[6], const Node &a, const Node &b, float fresh[6], uint32_t &mask) {
    float pa[6], pb[6];
    planes_of(a, pa);
    planes_of(b, pb);
    mask = 0u;
    for (int k = 0; k < 6; ++k) {
        const uint32_t want = __float_as_uint(parent[k]);
        if (__float_as_uint(pa[k]) == want) {
            mask |= 1u << k;
            fresh[k] = pb[k];
        } else if (__float_as_uint(pb[k]) == want) {
            fresh[k] = pa[k];
        } else {
            return false;
        }
    }
    return true;
}

__global__ __launch_bounds__(kThreads) void k_chain(const Node *__restrict__ nodes, uint32_t n_nodes, ChainNode *__restrict__ chain) {
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n_nodes; i += gridDim.x * kThreads) {
        if (nodes[i].leaf != kInterior) continue;
        const uint32_t l = i + 1, r = nodes[l].skip;
        const Node a = nodes[l], b = nodes[r];
        ChainNode w{};
        w.lx[0] = a.xlo, w.lx[1] = a.xhi, w.ly[0] = a.ylo, w.ly[1] = a.yhi, w.lz[0] = a.zlo, w.lz[1] = a.zhi;
        w.rx[0] = b.xlo, w.rx[1] = b.xhi, w.ry[0] = b.ylo, w.ry[1] = b.yhi, w.rz[0] = b.zlo, w.rz[1] = b.zhi;
        w.left = child_ref(nodes, l);
        w.right = child_ref(nodes, r);
        uint32_t below = 0u, mask1 = 0u, mask2 = 0u;
        float parent[6], fresh[6];
        if (a.leaf == kInterior) { // the children of L, told from L's box
            const uint32_t ll = l + 1, lr = nodes[ll].skip;
            planes_of(a, parent);
            if (encode_pair(parent, nodes[ll], nodes[lr], fresh, mask1)) {
                below |= 1u;
                w.n1xy[0] = fresh[0], w.n1xy[1] = fresh[1], w.n1xy[2] = fresh[2], w.n1xy[3] = fresh[3];
                w.n1z[0] = fresh[4], w.n1z[1] = fresh[5];
                w.ll = child_ref(nodes, ll);
                w.lr = child_ref(nodes, lr);
            }
        }
        if (b.leaf == kInterior) { // the children of R, told from R's box
            const uint32_t rl = r + 1, rr = nodes[rl].skip;
            planes_of(b, parent);
            if (encode_pair(parent, nodes[rl], nodes[rr], fresh, mask2)) {
                below |= 2u;
                w.n2xy[0] = fresh[0], w.n2xy[1] = fresh[1], w.n2xy[2] = fresh[2], w.n2xy[3] = fresh[3];
                w.n2z[0] = fresh[4], w.n2z[1] = fresh[5];
                w.rl = child_ref(nodes, rl);
                w.rr = child_ref(nodes, rr);
            }
        }
        w.meta = mask1 | (mask2 << 8) | (below << 16);
        chain[i - leaves_before(nodes, i)] = w;
    }
}

} // namespace

void build_chain_nodes(hipStream_t stream, const Node *d_nodes, uint32_t n_nodes, ChainNode *d_chain) {
    const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)n_nodes + kThreads - 1) / kThreads, 1u << 16);
    hipLaunchKernelGGL(k_chain, dim3(grid), dim3(kThreads), 0, stream, d_nodes, n_nodes, d_chain);
}

int build_bvh_device(hipStream_t stream, const BuildBox *h_boxes, uint32_t n, uint32_t sphere_begin, uint32_t sphere_end,
                     Node *d_nodes, uint32_t *h_order, uint32_t *depth, std::string &err) {
    static_assert(sizeof(BuildBox) == sizeof(DBox), "box layouts must agree");
    if (n == 0) {
        err = "empty object list";
        return RT_ERR_INVALID_ARGUMENT;
    }
    Scratch mem;
    Level lv{};
    DBox *d_boxes = nullptr;
    unsigned long long *keys[2] = {};
    uint32_t *stay[2] = {}, *ids[2] = {}, *heads = nullptr, *n_heads = nullptr;
    bool ok = mem.alloc(&d_boxes, n) && mem.alloc(&lv.order, n) && mem.alloc(&lv.seg_lo, n) && mem.alloc(&lv.seg_len, n) &&
              mem.alloc(&lv.seg_at, n) && mem.alloc(&lv.axis, n) && mem.alloc(&keys[0], n) && mem.alloc(&keys[1], n) &&
              mem.alloc(&stay[0], n) && mem.alloc(&stay[1], n) && mem.alloc(&ids[0], n) && mem.alloc(&ids[1], n) &&
              mem.alloc(&heads, n / kLongSegment + 2) && mem.alloc(&n_heads, 1);
    if (!ok) {
        err = "out of device memory while building the BVH";
        return RT_ERR_OUT_OF_MEMORY;
    }
    lv.boxes = d_boxes;
    lv.nodes = d_nodes;
    lv.n = n;
    lv.sphere_begin = sphere_begin;
    lv.sphere_end = sphere_end;
    BUILD_TRY(hipMemcpyAsync(d_boxes, h_boxes, (size_t)n * sizeof(DBox), hipMemcpyHostToDevice, stream));

    // one temporary buffer that fits both sorts
    uint32_t id_bits = 1;
    while (id_bits < 32 && (1ull << id_bits) < n) ++id_bits;
    size_t temp_a = 0, temp_b = 0;
    BUILD_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_a, keys[0], keys[1], stay[0], stay[1], (int)n, 0, 64, stream));
    BUILD_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_b, stay[1], stay[0], ids[0], ids[1], (int)n, 0, (int)id_bits, stream));
    const size_t temp_bytes = temp_a > temp_b ? temp_a : temp_b;
    uint8_t *temp = nullptr;
    if (!mem.alloc(&temp, temp_bytes)) {
        err = "out of device memory while building the BVH";
        return RT_ERR_OUT_OF_MEMORY;
    }

    const uint32_t grid = (n + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k_init, dim3(grid), dim3(kThreads), 0, stream, lv);
    uint32_t levels = 1;
    for (uint64_t longest = n; longest > 1; longest = (longest + 1) / 2, ++levels) {
        // segment lengths of a level differ by at most one, so a level is long or short as a whole
        if (longest / 2 >= kLongSegment) {
            BUILD_TRY(hipMemsetAsync(n_heads, 0, sizeof(uint32_t), stream));
            hipLaunchKernelGGL(k_collect_heads, dim3(grid), dim3(kThreads), 0, stream, lv, heads, n_heads);
            const uint32_t max_heads = (uint32_t)(n / (longest / 2)) + 1u;
            hipLaunchKernelGGL(k_nodes_long, dim3(max_heads), dim3(kThreads), 0, stream, lv, heads, n_heads);
        } else {
            hipLaunchKernelGGL(k_nodes_short, dim3(grid), dim3(kThreads), 0, stream, lv);
        }
        hipLaunchKernelGGL(k_keys, dim3(grid), dim3(kThreads), 0, stream, lv, keys[0], stay[0]);
        size_t tb = temp_bytes; // (descending key | ID), carrying the segment start along
        BUILD_TRY(hipcub::DeviceRadixSort::SortPairs(temp, tb, keys[0], keys[1], stay[0], stay[1], (int)n, 0, 64, stream));
        hipLaunchKernelGGL(k_low_words, dim3(grid), dim3(kThreads), 0, stream, keys[1], ids[0], n);
        tb = temp_bytes; // stable by segment start: every segment returns to its place, sorted
        BUILD_TRY(hipcub::DeviceRadixSort::SortPairs(temp, tb, stay[1], stay[0], ids[0], ids[1], (int)n, 0, (int)id_bits, stream));
        hipLaunchKernelGGL(k_split, dim3(grid), dim3(kThreads), 0, stream, lv, ids[1]);
    }
    hipLaunchKernelGGL(k_nodes_short, dim3(grid), dim3(kThreads), 0, stream, lv); // leaves of the last level
    BUILD_TRY(hipGetLastError());
    BUILD_TRY(hipMemcpyAsync(h_order, lv.order, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    BUILD_TRY(hipStreamSynchronize(stream));
    *depth = levels;
    return RT_OK;
}

} // namespace rt
