// kernels.hip — the wavefront path tracer for gfx950: generate / extend / shade / accumulate, the per-bounce
// fusion of the first three (k_bounce) and the whole bounce loop of a batch in one launch (k_paths).
//
// Every kernel loops over a queue of path slots whose length lives in device memory, so the host never has to
// read a count back between bounces.  One lane owns one path.  All arithmetic is binary32 in the reference's
// operation order (built with -ffp-contract=off), including its inf/NaN behaviour.
#include <hip/hip_runtime.h>

#include <mutex>
#include <type_traits>

#include "kernels.h"
#include "rt_math.h"

#ifndef RT_SERVICE_DIV
#define RT_SERVICE_DIV 4u
#endif
#ifndef RT_LEAF_DIV
#define RT_LEAF_DIV 4u
#endif

namespace rt {

#ifdef RT_LANE_STATS
// Tuning aid (variant builds only): lane-rounds of k_paths_lanes by what the lane was doing.
// 0 node rounds, 1 lanes stepping in them, 2 leaf rounds, 3 lanes stepping in them, 4 lanes finished (awaiting service),
// 5 lanes without a path, 6 lanes waiting at a leaf during a node round, 7 services, 8 lanes shaded in them
__device__ unsigned long long g_lane_stats[16];
extern "C" __attribute__((visibility("default"))) int rt_debug_lane_stats(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lane_stats), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        unsigned long long zero[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_lane_stats), zero, sizeof zero) != hipSuccess) return -1;
    }
    return 0;
}
#endif

namespace {

constexpr int kBlock = 256;
// ray_o.w of an in-flight path: index of its next random draw in the low kDrawBits bits, NaN flags above.
constexpr uint32_t kDrawBits = 28u;
constexpr uint32_t kDrawMask = (1u << kDrawBits) - 1u;

// The wave's mask of lanes where `pred` holds, straight from the compare (HIP's __ballot takes an int, which makes
// the compiler materialise the predicate in a register and compare it again).
__device__ __forceinline__ unsigned long long ballot(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }

// A pointer every lane holds the same value of, as a value the compiler knows to be uniform (so that it lives in
// scalar registers, which the scalar-base form of a global load needs).
template <class T> __device__ __forceinline__ T *uniform_pointer(T *p) {
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<T *>((uintptr_t)(((unsigned long long)hi << 32) | lo));
}

// The LDS address (byte offset within the workgroup's LDS) of an object in shared memory.
__device__ __forceinline__ uint32_t lds_address(const void *shared) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)shared;
}

__device__ __forceinline__ V3 ld3(const float *p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 xyz(float4 v) { return V3{v.x, v.y, v.z}; }
__device__ __forceinline__ float4 mk4(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }

// ------------------------------------------------------------------------------------------------------
// intersection
// ------------------------------------------------------------------------------------------------------

// Aabb::hit (src/aabb.rs:95-127) on a node stored as (lo.x, hi.x, lo.y, hi.y), (lo.z, hi.z, skip, leaf).  The
// reference recomputes 1/d per node; the value is the same every time, so it is computed once per ray.  Per
// axis it orders (t0, t1) with `t0 < t1` and then narrows [lo, hi] by compare-and-assign, which leaves the
// interval untouched by a NaN; v_max_f32 / v_min_f32 have exactly that behaviour because lo and hi themselves
// are never NaN.  The per-axis early `max <= min` exits are equivalent to one test at the end because the
// interval only ever shrinks.  The two planes of an axis sit in one register pair, the operand shape of the
// packed binary32 subtract and multiply; operations and their order per lane are the reference's.
typedef float v2f __attribute__((ext_vector_type(2)));

// v_max_f32 / v_min_f32 as the hardware does them (the other operand when one is a NaN).  fmaxf / fminf compile to
// the same instruction preceded by a quieting `v_max x, x` whenever the compiler cannot prove an operand is not a
// signalling NaN, which it cannot for the loop-carried closest distance; that extra instruction changes nothing.
__device__ __forceinline__ float max_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// A copy of `x` that lives in a vector register (so that a loop-invariant scalar is not moved there anew in
// every iteration in front of max_raw).
__device__ __forceinline__ float in_vgpr(float x) {
    float r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ float min_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// The three-operand forms: max(max(a, b), c) / min(min(a, b), c) with the same treatment of NaNs, in one instruction.
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float min3_raw(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// A ray as the slab test wants it: per axis the origin twice (subtracted from the low and from the high plane) and the
// reciprocal of the direction.
//
// Aabb::hit orders an axis' two distances with `t0 < t1` and narrows [lo, hi] by compare-and-assign.  For ordered
// operands that is lo = max(lo, min(t0, t1)), hi = min(hi, max(t0, t1)), which v_min_f32 / v_max_f32 compute in two
// instructions instead of a compare and two selects (the sign of a zero may differ, which no later comparison sees).
// The two forms part only when exactly ONE of t0, t1 is a NaN (the hardware min / max then return the other operand
// for both, compare-and-assign leaves one bound alone).  With plane - origin finite (any scene whose coordinates stay
// below 1.7e38 in magnitude) a lone NaN is 0 * inf: the direction component is a zero and the origin lies exactly on
// one of the two planes.  For +0 both forms reject the box.  For -0 they differ: compare-and-assign passes the axis
// iff low <= origin <= high (planes included), min / max would reject an origin on a plane.  Such rays are common (the
// diffuse sampler's degenerate samples have z = 0, negated for half of them), so they are not left to a fallback:
// for an axis whose direction is -0 the ray is stored with the reciprocal +inf and the origin replaced by the next
// binary32 above it for the low plane and the next below it for the high plane.  Then low - up(origin) is negative
// exactly when low <= origin, high - down(origin) positive exactly when origin <= high, and the min / max form with
// +inf passes the axis for exactly the origins compare-and-assign passes with -inf, and rejects the box for all others
// (origins one step outside give 0 * inf = NaN on one side and the rejecting infinity on the other).  NaN and infinite
// origins behave as before (up(+inf) = +inf, down(-inf) = -inf, NaN stays NaN).
struct Slabs {
    v2f ox, oy, oz;
    V3 inv;
};

__device__ __forceinline__ float next_up(float x) { // the next binary32 towards +inf; NaN and +inf stay
    const uint32_t u = f2u(x);
    if (x != x || u == 0x7F800000u) return x;
    if ((u << 1) == 0u) return u2f(1u);
    return u2f((u >> 31) ? u - 1u : u + 1u);
}
__device__ __forceinline__ float next_down(float x) { return -next_up(-x); }

// (The substitution is needed by few rays, and hardly ever on x or y; a wave skips an axis none of its lanes needs.)
__device__ __forceinline__ Slabs make_slabs(V3 o, V3 d) {
    Slabs s;
    s.inv = V3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    s.ox = v2f{o.x, o.x};
    s.oy = v2f{o.y, o.y};
    s.oz = v2f{o.z, o.z};
    const float inf = __builtin_inff();
    const bool nx = f2u(d.x) == 0x80000000u, ny = f2u(d.y) == 0x80000000u, nz = f2u(d.z) == 0x80000000u;
    if (nx) s.ox = v2f{next_up(o.x), next_down(o.x)}, s.inv.x = inf;
    if (ny) s.oy = v2f{next_up(o.y), next_down(o.y)}, s.inv.y = inf;
    if (nz) s.oz = v2f{next_up(o.z), next_down(o.z)}, s.inv.z = inf;
    return s;
}

// Aabb::hit as written (compare-and-assign, no preparation of the ray): for a single test of a box.
__device__ __forceinline__ bool box_hit_once(float4 a, float4 b, V3 o, V3 d, float lo, float hi) {
    const V3 inv = V3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    v2f tx = (v2f{a.x, a.y} - v2f{o.x, o.x}) * v2f{inv.x, inv.x};
    v2f ty = (v2f{a.z, a.w} - v2f{o.y, o.y}) * v2f{inv.y, inv.y};
    v2f tz = (v2f{b.x, b.y} - v2f{o.z, o.z}) * v2f{inv.z, inv.z};
    bool s = tx.x < tx.y;
    lo = max_raw(lo, s ? tx.x : tx.y);
    hi = min_raw(hi, s ? tx.y : tx.x);
    s = ty.x < ty.y;
    lo = max_raw(lo, s ? ty.x : ty.y);
    hi = min_raw(hi, s ? ty.y : ty.x);
    s = tz.x < tz.y;
    lo = max_raw(lo, s ? tz.x : tz.y);
    hi = min_raw(hi, s ? tz.y : tz.x);
    return !(hi <= lo);
}

// Aabb::hit (see Slabs).  `lo` and `hi` are never NaN.
__device__ __forceinline__ bool box_hit(float4 a, float4 b, const Slabs &r, float lo, float hi) {
    v2f tx = (v2f{a.x, a.y} - r.ox) * v2f{r.inv.x, r.inv.x};
    v2f ty = (v2f{a.z, a.w} - r.oy) * v2f{r.inv.y, r.inv.y};
    v2f tz = (v2f{b.x, b.y} - r.oz) * v2f{r.inv.z, r.inv.z};
    lo = max3_raw(lo, min_raw(tx.x, tx.y), min_raw(ty.x, ty.y));
    hi = min3_raw(hi, max_raw(tx.x, tx.y), max_raw(ty.x, ty.y));
    lo = max_raw(lo, min_raw(tz.x, tz.y));
    hi = min_raw(hi, max_raw(tz.x, tz.y));
    return !(hi <= lo);
}

struct Closest {
    float t;
    uint32_t slot;
    float u, v;
};

// TriangleRef::hit up to the range test (src/object/triangle_mesh.rs:140-170): Möller–Trumbore without
// culling, `contains` is inclusive at both ends.
__device__ __forceinline__ void tri_test(float4 a, float4 b, float4 c, uint32_t slot, V3 o, V3 d, float tmin, Closest &best) {
    V3 v0 = xyz(a), e1 = xyz(b), e2 = xyz(c);
    V3 p = cross(d, e2);
    float det = dot(e1, p);
    if (fabsf(det) < 1.1920929e-07f) return;
    float inv_det = 1.0f / det;
    V3 tv = o - v0;
    float u = dot(tv, p) * inv_det;
    if (u < 0.0f || u > 1.0f) return;
    V3 q = cross(tv, e1);
    float v = dot(d, q) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return;
    float t = dot(e2, q) * inv_det;
    if (!(tmin <= t && t <= best.t)) return;
    best.t = t;
    best.slot = slot;
    best.u = u;
    best.v = v;
}

__device__ __forceinline__ void tri_test(const float4 *__restrict__ prim, uint32_t slot, V3 o, V3 d, float tmin,
                                         Closest &best) {
    tri_test(prim[0], prim[1], prim[2], slot, o, d, tmin, best);
}

// Sphere::hit up to the range test (src/object/sphere.rs:55-72): `surrounds` is exclusive.
__device__ __forceinline__ void sphere_test(float4 a, uint32_t slot, V3 o, V3 d, float tmin, Closest &best) {
    V3 oc = xyz(a) - o;
    float aa = length_squared(d);
    float h = dot(d, oc);
    float c = length_squared(oc) - a.w * a.w;
    float discriminant = h * h - aa * c;
    if (discriminant < 0.0f) return;
    float sqrtd = sqrtf(discriminant);
    float root = (h - sqrtd) / aa;
    if (!(tmin < root && root < best.t)) {
        root = (h + sqrtd) / aa;
        if (!(tmin < root && root < best.t)) return;
    }
    best.t = root;
    best.slot = slot | kSphereBit;
    best.u = 0.0f;
    best.v = 0.0f;
}

__device__ __forceinline__ void sphere_test(const float4 *__restrict__ prim, uint32_t slot, V3 o, V3 d, float tmin,
                                            Closest &best) {
    sphere_test(prim[0], slot, o, d, tmin, best);
}

// BvhNode::hit (src/bvh.rs:295-317) as a stackless pre-order walk over the node table in device memory.  `best.t`
// plays the role of the shrinking `hit_range.max`: the reference hands the right sibling max = the left hit's
// distance, and every later node therefore sees the closest distance found so far in visiting order.
__device__ __forceinline__ Closest traverse(const float4 *__restrict__ nodes, const float4 *__restrict__ prims, uint32_t n_nodes, V3 o,
                                            V3 d, float tmin, float tmax) {
    const Slabs slabs = make_slabs(o, d);
    Closest best{tmax, kMissSlot, 0.0f, 0.0f};
    tmin = in_vgpr(tmin);
    uint32_t idx = 0;
    while (idx < n_nodes) {
        const float4 nlo = nodes[2 * idx], nhi = nodes[2 * idx + 1];
        const bool hit = box_hit(nlo, nhi, slabs, tmin, best.t);
        const uint32_t leaf = f2u(nhi.w);
        if (hit && leaf != kInterior) {
            const uint32_t slot = leaf & ~kSphereBit;
            if (leaf & kSphereBit)
                sphere_test(prims + 3 * (size_t)slot, slot, o, d, tmin, best);
            else
                tri_test(prims + 3 * (size_t)slot, slot, o, d, tmin, best);
        }
        idx = hit ? idx + 1u : f2u(nhi.z);
    }
    return best;
}

// ------------------------------------------------------------------------------------------------------
// The walk over tables in LDS with the slabs ordered by the sign of the direction.
//
// For a positive reciprocal the distance to an axis' low plane is the smaller of the two (subtraction and
// multiplication round monotonically and low <= high), for a negative one the distance to the high plane: which of the
// two distances is the entry is known per ray, and no instruction has to order them per box.  The LDS copy stores an
// axis as (low, high, low), so that the pair read at byte offset 0 is (low, high) and the pair at offset 4 is (high,
// low): a per-ray offset per axis delivers (entry plane, exit plane) directly, and the interval update is
// lo = max(lo, entries), hi = min(hi, exits) with NaN operands dropped, which is what Aabb::hit's compare-and-assign
// does with a NaN.  4 min / max per box test instead of 10 (or a compare and two selects per axis as written).
//
// Where that differs from Aabb::hit (src/aabb.rs:95-127): only when exactly one of an axis' two distances is a NaN,
// i.e. (for coordinates below 1.7e38) a direction component of +0 or -0 with the origin exactly on one of the planes.
// Aabb::hit then passes the axis for -0 iff low <= origin <= high and for +0 iff low < origin < high (a NaN makes
// `t0 < t1` false, and the else-branch assigns the infinity of the other plane to one bound or leaves both alone).
// Here both zeros get the reciprocal +inf, so entry = (low - o) * inf is -inf (no constraint) for low < o, NaN
// (dropped: no constraint) for low == o and +inf (reject) for low > o, and likewise the exit: that IS the rule for
// -0.  For +0 the origin is replaced by its lower binary32 neighbour for the low plane and its upper neighbour for the
// high plane, which turns `<=` into `<`: low - down(o) <= 0 iff low < o.  NaN and infinite origins behave as in
// Aabb::hit (both distances NaN: axis passed; both the same infinity: rejected).
// ------------------------------------------------------------------------------------------------------
constexpr uint32_t kSortedNodeBytes = 48u; // (xlo xhi xlo) (ylo yhi ylo) (zlo zhi zlo) skip leaf pad

struct SortedSlabs {
    v2f ox, oy, oz;        // origin for the (entry, exit) plane of each axis
    V3 inv;
    uint32_t ax, ay, az;   // byte offset of an axis' (entry, exit) pair within a node
};

__device__ __forceinline__ void sorted_axis(float o, float d, uint32_t base, v2f &origin, float &inv, uint32_t &at) {
    inv = 1.0f / d;
    origin = v2f{o, o};
    at = base;
    if (d == 0.0f) { // either zero
        inv = __builtin_inff();
        if (f2u(d) == 0u) origin = v2f{next_down(o), next_up(o)};
    } else if (inv < 0.0f) {
        at = base + 4u;
    }
}

__device__ __forceinline__ SortedSlabs make_sorted_slabs(V3 o, V3 d) {
    SortedSlabs s;
    sorted_axis(o.x, d.x, 0u, s.ox, s.inv.x, s.ax);
    sorted_axis(o.y, d.y, 12u, s.oy, s.inv.y, s.ay);
    sorted_axis(o.z, d.z, 24u, s.oz, s.inv.z, s.az);
    return s;
}

// BvhNode::hit as the stackless pre-order walk (traverse()) over the sorted LDS copy at LDS address `lds_nodes`.
template <bool COUNT = false>
__device__ __forceinline__ Closest traverse_sorted(uint32_t lds_nodes, const float4 *__restrict__ prims, uint32_t n_nodes, V3 o,
                                                   V3 d, float tmin, float tmax, uint32_t *visits = nullptr) {
    typedef __attribute__((address_space(3))) const float LdsF;
    typedef __attribute__((address_space(3))) const uint32_t LdsU;
    const SortedSlabs r = make_sorted_slabs(o, d);
    Closest best{tmax, kMissSlot, 0.0f, 0.0f};
    tmin = in_vgpr(tmin);
    uint32_t cur = lds_nodes; // LDS addresses throughout: the copy's skip links are absolute too
    const uint32_t end = lds_nodes + n_nodes * kSortedNodeBytes;
    while (cur < end) {
        LdsF *px = reinterpret_cast<LdsF *>((uintptr_t)(cur + r.ax));
        LdsF *py = reinterpret_cast<LdsF *>((uintptr_t)(cur + r.ay));
        LdsF *pz = reinterpret_cast<LdsF *>((uintptr_t)(cur + r.az));
        LdsU *links = reinterpret_cast<LdsU *>((uintptr_t)cur);
        const v2f tx = (v2f{px[0], px[1]} - r.ox) * v2f{r.inv.x, r.inv.x};
        const v2f ty = (v2f{py[0], py[1]} - r.oy) * v2f{r.inv.y, r.inv.y};
        const v2f tz = (v2f{pz[0], pz[1]} - r.oz) * v2f{r.inv.z, r.inv.z};
        const uint32_t skip = links[9], leaf = links[10];
        const float lo = max_raw(max3_raw(tmin, tx.x, ty.x), tz.x);
        const float hi = min_raw(min3_raw(best.t, tx.y, ty.y), tz.y);
        const bool hit = !(hi <= lo);
        if (COUNT) ++visits[0];
        if (hit && leaf != kInterior) {
            const uint32_t slot = leaf & ~kSphereBit;
            if (COUNT) ++visits[(leaf & kSphereBit) ? 2 : 1];
            if (leaf & kSphereBit)
                sphere_test(prims + 3 * (size_t)slot, slot, o, d, tmin, best);
            else
                tri_test(prims + 3 * (size_t)slot, slot, o, d, tmin, best);
        }
        cur = hit ? cur + kSortedNodeBytes : skip;
    }
    return best;
}

// The same walk for scenes of a few dozen nodes: the LDS copy holds every node eight times, once per combination of
// direction signs, each copy with its planes already in (entry, exit) order and with its own skip link, all as absolute
// LDS addresses.  A ray picks its copy once (32 bytes per copy, 256 per node) and a box test needs no address
// arithmetic at all: two 16-byte reads, six packed operations, four min / max.
constexpr uint32_t kOctantNodeBytes = 256u;

template <bool COUNT = false>
__device__ __forceinline__ Closest traverse_octants(uint32_t lds_nodes, const float4 *__restrict__ prims, uint32_t n_nodes, V3 o, V3 d,
                                                    float tmin, float tmax, uint32_t *visits = nullptr) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const v4f LdsF4;
    const SortedSlabs r = make_sorted_slabs(o, d);
    Closest best{tmax, kMissSlot, 0.0f, 0.0f};
    tmin = in_vgpr(tmin);
    // the copy for this ray's signs: x reversed -> + 32, y -> + 64, z -> + 128 (ax, ay, az are 4 above their axis' base then)
    uint32_t cur = lds_nodes + (r.ax << 3) + ((r.ay - 12u) << 4) + ((r.az - 24u) << 5);
    const uint32_t end = lds_nodes + n_nodes * kOctantNodeBytes;
    while (cur < end) {
        LdsF4 *node = reinterpret_cast<LdsF4 *>((uintptr_t)cur);
        const v4f a = node[0], b = node[1];
        const v2f tx = (v2f{a.x, a.y} - r.ox) * v2f{r.inv.x, r.inv.x};
        const v2f ty = (v2f{a.z, a.w} - r.oy) * v2f{r.inv.y, r.inv.y};
        const v2f tz = (v2f{b.x, b.y} - r.oz) * v2f{r.inv.z, r.inv.z};
        const uint32_t skip = f2u(b.z), leaf = f2u(b.w);
        const float lo = max_raw(max3_raw(tmin, tx.x, ty.x), tz.x);
        const float hi = min_raw(min3_raw(best.t, tx.y, ty.y), tz.y);
        const bool hit = !(hi <= lo);
        if (COUNT) ++visits[0];
        if (hit && leaf != kInterior) {
            const uint32_t slot = leaf & ~kSphereBit;
            if (COUNT) ++visits[(leaf & kSphereBit) ? 2 : 1];
            if (leaf & kSphereBit)
                sphere_test(prims + 3 * (size_t)slot, slot, o, d, tmin, best);
            else
                tri_test(prims + 3 * (size_t)slot, slot, o, d, tmin, best);
        }
        cur = hit ? cur + kOctantNodeBytes : skip;
    }
    return best;
}

// World::hit (src/object/world.rs:21-39): every object in list order with a shrinking max.
__device__ __forceinline__ Closest scan_world(const SceneDev &sc, V3 o, V3 d, float tmin, float tmax) {
    const float4 *__restrict__ prims = reinterpret_cast<const float4 *>(sc.prims);
    Closest best{tmax, kMissSlot, 0.0f, 0.0f};
    for (uint32_t id = 0; id < sc.n_prims; ++id) {
        uint32_t leaf = sc.slot_of_prim[id];
        uint32_t slot = leaf & ~kSphereBit;
        if (leaf & kSphereBit)
            sphere_test(prims + 3 * (size_t)slot, slot, o, d, tmin, best);
        else
            tri_test(prims + 3 * (size_t)slot, slot, o, d, tmin, best);
    }
    return best;
}

// ------------------------------------------------------------------------------------------------------
// BvhNode::hit with the child boxes stored in the parent (ChainNode, rt_device.h) and a per-lane stack.
//
// Aabb::hit narrows [lo, hi] = [t_min, closest so far] by the three slabs and reports a hit unless hi <= lo.
// The slabs' contribution does not depend on the closest distance: with E = max(t_min, the three entry
// distances) and X = min(the three exit distances) (NaNs dropped as the compare-and-assign form drops them) the
// box is hit iff !(min(closest, X) <= E).  So both children's (E, X) are computed when their parent is entered:
// the left child is decided at once (nothing happens between the parent's and its visit), the right child is
// pushed as (E, reference) if it passes with the present closest distance, and decided again when it is popped,
// with the closest distance of that moment, which is exactly when and with what BvhNode::hit tests it.  Since the
// closest distance only shrinks, a right child that fails at push time would fail later too, and one that passed
// (so X > E) passes later iff closest > E: the stack entry needs E only.  Visiting order, primitive tests and
// every comparison are the reference's; what changes is that a node whose box is missed costs no memory access
// and one 128-byte read serves a node's children and grandchildren (step_node below).
//
// Why 128-byte records: a scattered read costs the memory system one 128-byte line whatever part of it is used
// (measured: 5.4e10 reads/s of 32 bytes, 5.3e10 of 64, 5.0e10 of 128, 2.3e10 of 256 from a 2 GB table), and with a
// few thousand independent walks per L2 only the top ten levels of a large tree stay cached, so the number of
// lines fetched per ray is what bounds a scene read from HBM.
//
// The stack: the K most recent entries live in LDS (entry e of a thread at lds[(e % K) * THREADS]), older ones
// are moved to `spill` in device memory and come back when the LDS part runs empty.  K must be a power of two (the
// ring is indexed with `& (K - 1)`) and K * THREADS * 8 bytes of LDS per block must leave room for the blocks a CU is to
// hold.  A step descends one or two levels and leaves at most two entries behind per level it descends (an L that is
// not held plus a held (RL, RR) pair: two entries for one level), so at most 2 * depth entries exist at a time; the spill
// area has room for 2 * depth + 2.
// ------------------------------------------------------------------------------------------------------
// The 128 bytes of a chain record for every active lane: from the block's LDS copy of the top records when `in_top`
// (at LDS address `lds_at`), else from device memory at `p`; both kinds of loads issued under their own execution
// mask and awaited together (see load_walk_node).
__device__ __forceinline__ void load_chain_record(const float4 *__restrict__ p, uint32_t lds_at, bool in_top, float4 out[8]) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f r0, r1, r2, r3, r4, r5, r6, r7;
    unsigned long long saved;
    const unsigned long long top = ballot(in_top);
    asm volatile("s_and_saveexec_b64 %[saved], %[top]\n\t"
                 "ds_read_b128 %[r0], %[la]\n\t"
                 "ds_read_b128 %[r1], %[la] offset:16\n\t"
                 "ds_read_b128 %[r2], %[la] offset:32\n\t"
                 "ds_read_b128 %[r3], %[la] offset:48\n\t"
                 "ds_read_b128 %[r4], %[la] offset:64\n\t"
                 "ds_read_b128 %[r5], %[la] offset:80\n\t"
                 "ds_read_b128 %[r6], %[la] offset:96\n\t"
                 "ds_read_b128 %[r7], %[la] offset:112\n\t"
                 "s_andn2_b64 exec, %[saved], %[top]\n\t"
                 "global_load_dwordx4 %[r0], %[ga], off\n\t"
                 "global_load_dwordx4 %[r1], %[ga], off offset:16\n\t"
                 "global_load_dwordx4 %[r2], %[ga], off offset:32\n\t"
                 "global_load_dwordx4 %[r3], %[ga], off offset:48\n\t"
                 "global_load_dwordx4 %[r4], %[ga], off offset:64\n\t"
                 "global_load_dwordx4 %[r5], %[ga], off offset:80\n\t"
                 "global_load_dwordx4 %[r6], %[ga], off offset:96\n\t"
                 "global_load_dwordx4 %[r7], %[ga], off offset:112\n\t"
                 "s_mov_b64 exec, %[saved]\n\t"
                 "s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [r4] "=&v"(r4), [r5] "=&v"(r5), [r6] "=&v"(r6),
                   [r7] "=&v"(r7), [saved] "=&s"(saved)
                 : [top] "s"(top), [la] "v"(lds_at), [ga] "v"(p)
                 : "memory");
    const v4f r[8] = {r0, r1, r2, r3, r4, r5, r6, r7};
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = make_float4(r[i].x, r[i].y, r[i].z, r[i].w);
}

__device__ __forceinline__ void slab_pair(float xlo, float xhi, float ylo, float yhi, float zlo, float zhi, const Slabs &r,
                                          float tmin, float &E, float &X) {
    v2f tx = (v2f{xlo, xhi} - r.ox) * v2f{r.inv.x, r.inv.x};
    v2f ty = (v2f{ylo, yhi} - r.oy) * v2f{r.inv.y, r.inv.y};
    v2f tz = (v2f{zlo, zhi} - r.oz) * v2f{r.inv.z, r.inv.z};
    // min / max drop a NaN operand, as compare-and-assign leaves a bound alone (see Slabs); an exit that is NaN on all
    // three axes stays NaN and is dropped by the caller's min(closest, X)
    E = max_raw(max3_raw(tmin, min_raw(tx.x, tx.y), min_raw(ty.x, ty.y)), min_raw(tz.x, tz.y));
    X = min3_raw(max_raw(tx.x, tx.y), max_raw(ty.x, ty.y), max_raw(tz.x, tz.y));
}

template <int K, int THREADS>
struct LaneStack {
    static_assert(K > 0 && (K & (K - 1)) == 0, "the LDS part of the stack is a ring indexed with & (K - 1)");
    uint2 *lds;    // this thread's column of the block's LDS stack area
    uint2 *spill;  // this thread's column of the spill area
    uint32_t stride;
    uint32_t sp = 0, lo = 0; // entries [lo, sp) are in LDS, [0, lo) in the spill area
    __device__ __forceinline__ void push(float e, uint32_t ref) {
        if (sp - lo == (uint32_t)K) {
            spill[(size_t)lo * stride] = lds[(lo & (K - 1)) * THREADS];
            ++lo;
        }
        lds[(sp & (K - 1)) * THREADS] = make_uint2(f2u(e), ref);
        ++sp;
    }
    __device__ __forceinline__ uint2 pop() {
        --sp;
        if (sp < lo) {
            lo = sp;
            return spill[(size_t)sp * stride];
        }
        return lds[(sp & (K - 1)) * THREADS];
    }
};

// One ray's walk: begin() tests the root box (BvhNode::hit on the root), step() handles the node or primitive
// `cur` names and moves to the next one in the reference's order; the walk is over when cur == kRefNone.
template <int K, int THREADS>
struct StackWalk {
    LaneStack<K, THREADS> stack;
    V3 o, d;
    Slabs slabs;
    float tmin;
    Closest best;
    uint32_t cur;

    __device__ __forceinline__ void init(uint2 *lds_stack, const StackSpill &spill) {
        stack.lds = lds_stack + threadIdx.x;
        stack.spill = spill.spill + (size_t)blockIdx.x * THREADS + threadIdx.x;
        stack.stride = spill.stride;
        cur = kRefNone;
    }
    __device__ __forceinline__ void begin(float4 root_lo, float4 root_hi, uint32_t root_ref, V3 origin, V3 dir, float t_min, float t_max) {
        o = origin;
        d = dir;
        slabs = make_slabs(o, d);
        best = Closest{t_max, kMissSlot, 0.0f, 0.0f};
        tmin = in_vgpr(t_min);
        float E, X;
        slab_pair(root_lo.x, root_lo.y, root_lo.z, root_lo.w, root_hi.x, root_hi.y, slabs, tmin, E, X);
        cur = !(min_raw(best.t, X) <= E) ? root_ref : kRefNone;
    }
    // One pair of children of a node with planes `parent` (xlo, xhi, ylo, yhi, zlo, zhi): plane k of the left
    // child is the parent's when bit k of `mask` is set and `fresh[k]` otherwise, the right child takes the other.
    // Decides both boxes with the present closest distance; el, er are their entry distances.
    __device__ __forceinline__ void pair_below(const float parent[6], const float fresh[6], uint32_t mask, bool &hit_l, bool &hit_r,
                                               float &el, float &er) {
        float left[6], right[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const bool inherit = (mask >> k) & 1u;
            right[k] = inherit ? fresh[k] : parent[k];
            left[k] = inherit ? parent[k] : fresh[k];
        }
        float xl, xr;
        slab_pair(left[0], left[1], left[2], left[3], left[4], left[5], slabs, tmin, el, xl);
        slab_pair(right[0], right[1], right[2], right[3], right[4], right[5], slabs, tmin, er, xr);
        hit_l = !(min_raw(best.t, xl) <= el);
        hit_r = !(min_raw(best.t, xr) <= er);
    }

    // After a step: `next` if the step named a node to go on with, else the most recent pending right child that
    // still passes with the present closest distance.
    __device__ __forceinline__ void advance(uint32_t next) {
        while (next == kRefNone && stack.sp > 0u) {
            const uint2 e = stack.pop();
            if (!(best.t <= u2f(e.x))) next = e.y;
        }
        cur = next;
    }
    // kRefLeaf is the sign bit and kRefNone is -1: a leaf reference is < -1, a node reference >= 0
    __device__ __forceinline__ bool at_leaf() const { return (int32_t)cur < -1; }
    __device__ __forceinline__ bool at_node() const { return (int32_t)cur >= 0; }

    // cur names a primitive: the 48 bytes of its record (read no further: a primitive record read as 64 bytes would
    // reach into the next cache line more often).  Measured and rejected: primitive records at a 64-byte stride,
    // which never straddle a line but pack fewer neighbouring leaves into one (10 M soup 0.44 -> 0.40 Grays/s), and
    // a left leaf's triangle carried in its parent's record (saves the fetch, but the extra triangle test in the
    // node step costs more issue slots than the fetch: 0.452 -> 0.436).
    __device__ __forceinline__ void step_leaf(const float4 *__restrict__ prims) {
        const uint32_t index = cur & kRefIndex;
        const float4 *__restrict__ p = prims + 3 * (size_t)index;
        const float4 a = p[0], b = p[1], c = p[2];
        if (cur & kRefSphere)
            sphere_test(a, index, o, d, tmin, best);
        else
            tri_test(a, b, c, index, o, d, tmin, best);
        advance(kRefNone);
    }

    // cur names an interior node X: the 128 bytes of its record decide X's children L, R and, where the record holds
    // them, the children of those that pass.  BvhNode::hit would visit, in this order, LL's subtree, LR's, RL's, RR's
    // (L's or R's own where the record does not hold their children): the first of them that passes is walked next,
    // the others are pushed in reverse order.  L's children are decided for good now (nothing happens between L's
    // test and theirs).  R's children wait on the stack while L's subtree is walked; BvhNode::hit tests R and then the
    // child at that later moment, and since R passed now (so its exit lies beyond its entry) and the child's own
    // exit-against-entry test is done now, what is left for later is "closest > R's entry and closest > the child's
    // entry": the entry pushed with the child is the larger of the two.
    // `lds_top`: LDS address of the block's copy of the top records (read when cur carries kRefTop); kRefNone when the
    // launch stages none.
    __device__ __forceinline__ void step_node(const float4 *__restrict__ chain, uint32_t lds_top = kRefNone) {
        float4 rec[8];
        if (lds_top == kRefNone) {
            const float4 *__restrict__ p = chain + 8 * (size_t)cur;
#pragma unroll
            for (int i = 0; i < 8; ++i) rec[i] = p[i];
        } else {
            const bool in_top = (cur & kRefTop) != 0u;
            const uint32_t at = cur & ~kRefTop;
            load_chain_record(chain + 8 * (size_t)at, lds_top + at * 128u, in_top, rec);
        }
        const float4 a = rec[0], b = rec[1], c = rec[2], q = rec[3], n1 = rec[4], nz = rec[5], n2 = rec[6], refs = rec[7];
        const uint32_t meta = f2u(q.z);
        float el, xl, er, xr;
        slab_pair(a.x, a.y, a.z, a.w, b.x, b.y, slabs, tmin, el, xl);
        slab_pair(c.x, c.y, c.z, c.w, b.z, b.w, slabs, tmin, er, xr);
        const bool hit_l = !(min_raw(best.t, xl) <= el);
        const bool hit_r = !(min_raw(best.t, xr) <= er);
        uint32_t r0 = kRefNone, r1 = kRefNone, r2 = kRefNone, r3 = kRefNone; // in visiting order
        float e1 = 0.0f, e2 = er, e3 = 0.0f;
        if (hit_l) {
            if (meta & (1u << 16)) {
                const float box[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
                const float fresh[6] = {n1.x, n1.y, n1.z, n1.w, nz.x, nz.y};
                bool h0, h1;
                float e0;
                pair_below(box, fresh, meta, h0, h1, e0, e1);
                if (h0) r0 = f2u(refs.x);
                if (h1) r1 = f2u(refs.y);
            } else {
                r0 = f2u(q.x);
            }
        }
        if (hit_r) {
            if (meta & (1u << 17)) {
                const float box[6] = {c.x, c.y, c.z, c.w, b.z, b.w};
                const float fresh[6] = {n2.x, n2.y, n2.z, n2.w, nz.z, nz.w};
                bool h2, h3;
                pair_below(box, fresh, meta >> 8, h2, h3, e2, e3);
                e2 = max_raw(e2, er);
                e3 = max_raw(e3, er);
                if (h2) r2 = f2u(refs.z);
                if (h3) r3 = f2u(refs.w);
            } else {
                r2 = f2u(q.y);
            }
        }
        const bool any0 = r0 != kRefNone, any01 = any0 || r1 != kRefNone, any012 = any01 || r2 != kRefNone;
        if (r3 != kRefNone && any012) stack.push(e3, r3);
        if (r2 != kRefNone && any01) stack.push(e2, r2);
        if (r1 != kRefNone && any0) stack.push(e1, r1);
        advance(any0 ? r0 : r1 != kRefNone ? r1 : r2 != kRefNone ? r2 : r3);
    }

    // One step whatever cur names (lanes of a wave in lockstep: one memory round per step).
    __device__ __forceinline__ void step(const float4 *__restrict__ chain, const float4 *__restrict__ prims) {
        if (cur & kRefLeaf)
            step_leaf(prims);
        else
            step_node(chain);
    }
};

// One record of the walk table for every active lane: record `at` (a byte offset) from the block's LDS copy of the
// treetop when at < top_bytes, else through the caches (scalar base + 32-bit offset).  `lds_top` is the LDS address (byte
// offset within the block's LDS) of the staged treetop.  Both kinds of loads are issued under their own execution
// mask and awaited together, so a round costs the longer of the two latencies once, and the four 16-byte halves land in
// the same two register quads (each lane is written by the one load that was issued for it).  Written in assembly
// because the compiler turns a two-sided load into generic (flat) loads and splits them.  `nodes` must be uniform.
__device__ __forceinline__ void load_walk_node(const float4 *__restrict__ nodes, uint32_t lds_top, uint32_t top_bytes, uint32_t at,
                                               float4 &nlo, float4 &nhi) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f lo4, hi4;
    unsigned long long saved;
    const unsigned long long in_top = ballot(at < top_bytes);
    asm volatile("s_and_saveexec_b64 %[saved], %[top]\n\t"
                 "ds_read_b128 %[lo], %[la]\n\t"
                 "ds_read_b128 %[hi], %[la] offset:16\n\t"
                 "s_andn2_b64 exec, %[saved], %[top]\n\t"
                 "global_load_dwordx4 %[lo], %[off], %[base]\n\t"
                 "global_load_dwordx4 %[hi], %[off], %[base] offset:16\n\t"
                 "s_mov_b64 exec, %[saved]\n\t"
                 "s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : [lo] "=&v"(lo4), [hi] "=&v"(hi4), [saved] "=&s"(saved)
                 : [top] "s"(in_top), [la] "v"(at + lds_top), [off] "v"(at), [base] "s"(nodes)
                 : "memory");
    nlo = make_float4(lo4.x, lo4.y, lo4.z, lo4.w);
    nhi = make_float4(hi4.x, hi4.y, hi4.z, hi4.w);
}

// The stackless pre-order walk (traverse() above) in the same begin / step form, for one lane per path over tables
// that sit in the caches, on the walk table (WalkNode, rt_device.h): cur is the byte offset of the node to test next,
// or kRefLeaf | slot when the node just tested was a leaf whose box is hit and its primitive test is still due (then
// `resume` is where to go on), or kRefNone when the walk is over.  Nodes below `top_bytes` are read from the block's
// LDS copy of the treetop, the others through the caches.
struct TopWalk {
    V3 o, d;
    Slabs slabs;
    float tmin;
    Closest best;
    uint32_t cur, resume;

    __device__ __forceinline__ void init(uint2 *, const StackSpill &) { cur = kRefNone; }
    __device__ __forceinline__ void begin(float4, float4, uint32_t, V3 origin, V3 dir, float t_min, float t_max) {
        o = origin;
        d = dir;
        slabs = make_slabs(o, d);
        best = Closest{t_max, kMissSlot, 0.0f, 0.0f};
        tmin = in_vgpr(t_min);
        cur = 0u; // the root's own box is the first record's
    }
    __device__ __forceinline__ bool at_leaf() const { return (int32_t)cur < -1; } // kRefLeaf is the sign bit, kRefNone is -1
    __device__ __forceinline__ bool at_node() const { return (int32_t)cur >= 0; }
    __device__ __forceinline__ void step_leaf(const float4 *__restrict__ prims) {
        const uint32_t slot = cur & kRefIndex;
        if (cur & kRefSphere)
            sphere_test(prims + 3 * (size_t)slot, slot, o, d, tmin, best);
        else
            tri_test(prims + 3 * (size_t)slot, slot, o, d, tmin, best);
        cur = resume;
    }
    __device__ __forceinline__ void step_node(const float4 *__restrict__ nodes, uint32_t lds_top, uint32_t top_bytes) {
        float4 nlo, nhi;
        load_walk_node(nodes, lds_top, top_bytes, cur, nlo, nhi);
        const bool hit = box_hit(nlo, nhi, slabs, tmin, best.t);
        const uint32_t on_hit = f2u(nhi.w), on_miss = f2u(nhi.z);
        if (hit && (on_hit & kRefLeaf)) resume = on_miss;
        cur = hit ? on_hit : on_miss;
    }
};

// The same walk for a wave whose lanes stay together (traverse() on the walk table): a lane at a hit leaf tests its
// primitive at once.
__device__ __forceinline__ Closest traverse_walk(const float4 *__restrict__ nodes, uint32_t lds_top, uint32_t top_bytes,
                                                 const float4 *__restrict__ prims, V3 o, V3 d, float tmin, float tmax) {
    const Slabs slabs = make_slabs(o, d);
    Closest best{tmax, kMissSlot, 0.0f, 0.0f};
    tmin = in_vgpr(tmin);
    uint32_t cur = 0u;
    while (cur != kRefNone) {
        float4 nlo, nhi;
        load_walk_node(nodes, lds_top, top_bytes, cur, nlo, nhi);
        const bool hit = box_hit(nlo, nhi, slabs, tmin, best.t);
        const uint32_t on_hit = f2u(nhi.w), on_miss = f2u(nhi.z);
        if (hit && (on_hit & kRefLeaf)) {
            const uint32_t slot = on_hit & kRefIndex;
            if (on_hit & kRefSphere)
                sphere_test(prims + 3 * (size_t)slot, slot, o, d, tmin, best);
            else
                tri_test(prims + 3 * (size_t)slot, slot, o, d, tmin, best);
            cur = on_miss;
        } else {
            cur = hit ? on_hit : on_miss;
        }
    }
    return best;
}

template <int K, int THREADS>
__device__ __forceinline__ Closest traverse_stack(const float4 *__restrict__ chain, const float4 *__restrict__ prims,
                                                  float4 root_lo, float4 root_hi, uint32_t root_ref, V3 o, V3 d, float tmin,
                                                  float tmax, uint2 *lds_stack, StackSpill spill) {
    StackWalk<K, THREADS> walk;
    walk.init(lds_stack, spill);
    walk.begin(root_lo, root_hi, root_ref, o, d, tmin, tmax);
    while (walk.cur != kRefNone) walk.step(chain, prims);
    return walk.best;
}

// ------------------------------------------------------------------------------------------------------
// surface reconstruction and materials
// ------------------------------------------------------------------------------------------------------
struct Surface {
    V3 point;
    V3 normal; // face-forwarded
    bool front_facing;
    float tu, tv; // texture coordinates
    uint32_t material;
    uint32_t prim_id;
};

// Sphere::get_uv (src/object/sphere.rs:42-52); acos / atan2 are the restatements of glibc's acosf / atan2f in
// rt_math.h, so the texture coordinates equal the host's bit for bit.
__device__ __forceinline__ void sphere_uv(V3 p, float &u, float &v) {
    const float PI = 3.14159265358979323846f;
    float theta = acos_f32(-p.y);
    float phi = atan2_f32(-p.z, p.x) + PI;
    u = phi / (2.0f * PI);
    v = theta / PI;
}

// What TriangleRef::hit / Sphere::hit compute after the range test plus HitRecord::new
// (src/object/triangle_mesh.rs:172-187, src/object/sphere.rs:73-82, src/object/mod.rs:39-62).
template <bool NEED_UV>
__device__ __forceinline__ Surface make_surface(const SceneDev &sc, const float4 *__restrict__ prims, V3 o, V3 d,
                                                float t, uint32_t leaf, float u, float v) {
    Surface s;
    uint32_t slot = leaf & ~kSphereBit;
    s.point = o + d * t;
    V3 outward;
    s.tu = 0.0f;
    s.tv = 0.0f;
    if (leaf & kSphereBit) {
        float4 a = prims[3 * (size_t)slot];
        float4 b = prims[3 * (size_t)slot + 1];
        outward = (s.point - xyz(a)) / a.w;
        s.prim_id = f2u(b.x);
        s.material = f2u(b.y);
        if (NEED_UV) sphere_uv(s.point, s.tu, s.tv);
    } else {
        float4 a = prims[3 * (size_t)slot];
        float4 b = prims[3 * (size_t)slot + 1];
        s.prim_id = f2u(a.w);
        s.material = f2u(b.w);
        const float4 *__restrict__ sh = reinterpret_cast<const float4 *>(sc.tri_shade + slot);
        float4 s0 = sh[0], s1 = sh[1], s2 = sh[2], s3 = sh[3];
        uint32_t flags = f2u(s3.w);
        V3 n0 = V3{s0.x, s0.y, s0.z};
        if (flags & kTriHasNormals) {
            V3 n1 = V3{s0.w, s1.x, s1.y};
            V3 n2 = V3{s1.z, s1.w, s2.x};
            outward = (n0 * (1.0f - u - v) + n1 * u) + n2 * v;
        } else {
            outward = n0;
        }
        if (NEED_UV && (flags & kTriHasUvs)) {
            float w = 1.0f - u - v;
            s.tu = s2.y * w + s2.w * u + s3.y * v;
            s.tv = s2.z * w + s3.x * u + s3.z * v;
        }
    }
    s.front_facing = dot(d, outward) < 0.0f;
    s.normal = s.front_facing ? outward : -outward;
    return s;
}

__device__ __forceinline__ long long f32_to_i64_sat(float f) { // Rust `as i64`
    if (f != f) return 0;
    if (f >= 9223372036854775808.0f) return 0x7FFFFFFFFFFFFFFFll;
    if (f <= -9223372036854775808.0f) return (long long)0x8000000000000000ull;
    return (long long)f;
}
__device__ __forceinline__ uint32_t f32_to_u32_sat(float f) { // Rust `as u32`
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
__device__ __forceinline__ float clamp01(float x) { // Range::UNIT.clamp (src/range.rs:44-52)
    if (x < 0.0f) return 0.0f;
    if (x > 1.0f) return 1.0f;
    return x;
}

// Texture::value_at (src/texture.rs:47-51,70-83,115-136).
__device__ V3 texture_value(const SceneDev &sc, int32_t ti, float tu, float tv, V3 p) {
    for (uint32_t guard = 0; guard <= sc.n_textures; ++guard) {
        const TextureRec &t = sc.textures[ti];
        if (t.kind == 0u) return ld3(t.rgb);
        if (t.kind == 1u) {
            long long x = f32_to_i64_sat(floorf(t.inv_scale * p.x));
            long long y = f32_to_i64_sat(floorf(t.inv_scale * p.y));
            long long z = f32_to_i64_sat(floorf(t.inv_scale * p.z));
            long long sum = (long long)((unsigned long long)x + (unsigned long long)y + (unsigned long long)z);
            ti = (sum % 2 == 0) ? t.even : t.odd;
            continue;
        }
        if (t.height == 0u || t.width == 0u) return V3{0.0f, 1.0f, 1.0f};
        float cu = clamp01(tu);
        float cv = 1.0f - clamp01(tv);
        uint32_t i = f32_to_u32_sat(cu * (float)t.width);
        i = min(i, t.height - 1u); // the reference clamps the column with the HEIGHT (src/texture.rs:125)
        i = min(i, t.width - 1u);  // where it would index past the row (a panic there) stay inside it
        uint32_t j = min(f32_to_u32_sat(cv * (float)t.height), t.height - 1u);
        const uint8_t *px = sc.texels + t.pixel_offset + 3ull * ((unsigned long long)j * t.width + i);
        const float color_scale = 1.0f / 255.0f;
        return V3{(float)px[0] * color_scale, (float)px[1] * color_scale, (float)px[2] * color_scale};
    }
    return V3{0.0f, 0.0f, 0.0f};
}

__device__ __forceinline__ V3 material_color(const SceneDev &sc, const MaterialRec &m, const Surface &s) {
    if (m.solid) return ld3(m.rgb);
    return texture_value(sc, m.texture, s.tu, s.tv, s.point);
}

// src/sample.rs:7-30 with src/math.rs:3-5: the concentric-disk sampler binds (radius, angle) to
// (theta, r); kept as written, so the returned vector can be longer than 1 and z collapses to 0.
__device__ __forceinline__ V3 cosine_hemisphere(float ux, float uy) {
    const float FRAC_PI_4 = 0.785398163397448309615660845819875721f;
    const float FRAC_PI_2 = 1.57079632679489661923132169163975144f;
    float ox = 2.0f * ux - 1.0f;
    float oy = 2.0f * uy - 1.0f;
    float dx, dy;
    if (ox == 0.0f && oy == 0.0f) {
        dx = 1.0f;
        dy = 1.0f;
    } else {
        float theta, r;
        if (fabsf(ox) > fabsf(oy)) {
            theta = ox;
            r = FRAC_PI_4 * (oy / ox);
        } else {
            theta = oy;
            r = FRAC_PI_2 - FRAC_PI_4 * (ox / oy);
        }
        dx = r * cos_unit(theta);
        dy = r * sin_unit(theta);
    }
    float z = sqrtf(fmaxf(1.0f - (dx * dx) - (dy * dy), 0.0f));
    return V3{dx, dy, z};
}

// src/vec3.rs:65-72,83-85
__device__ __forceinline__ V3 gen_unit_vector(Rng &rng) {
    for (;;) {
        float x = rng_random_range(rng, -1.0f, 1.0f);
        float y = rng_random_range(rng, -1.0f, 1.0f);
        float z = rng_random_range(rng, -1.0f, 1.0f);
        V3 p = V3{x, y, z};
        if (length_squared(p) < 1.0f) return normalize(p);
    }
}

// src/material.rs:116-121; powi(5) is x * ((x*x) * (x*x)).
__device__ __forceinline__ float reflectance(float cosine, float refraction_index) {
    float r0 = (1.0f - refraction_index) / (1.0f + refraction_index);
    r0 *= r0;
    float x = 1.0f - cosine;
    float x2 = x * x;
    float x5 = x * (x2 * x2);
    return r0 + (1.0f - r0) * x5;
}

struct Scatter {
    V3 attenuation;
    V3 direction;
    float pdf; // 1 when the reference's pdf is None (src/renderer.rs:53)
};

// Scatterable::scatter for the four leaf materials (src/material.rs:47-64,72-82,89-114,129-132).
__device__ __forceinline__ bool scatter_leaf(const SceneDev &sc, const MaterialRec &m, V3 d, const Surface &s, Rng &rng,
                                             Scatter &out) {
    const float FRAC_1_PI = 0.318309886183790671537767526745028724f;
    switch (m.kind) {
    case 0u: { // Lambertian: world-space sample, no shading frame
        uint32_t a, b;
        rng_next_2u32(rng, a, b);
        V3 w_i = cosine_hemisphere(u32_to_unit(a), u32_to_unit(b));
        if (-d.z < 0.0f) w_i.z *= -1.0f;
        out.direction = w_i;
        out.attenuation = material_color(sc, m, s) * FRAC_1_PI;
        out.pdf = fabsf(w_i.z) * FRAC_1_PI;
        return true;
    }
    case 1u: { // Metal
        V3 reflected = reflect(d, s.normal);
        V3 fuzz = gen_unit_vector(rng) * m.param;
        out.direction = normalize(reflected) + fuzz;
        out.attenuation = material_color(sc, m, s);
        out.pdf = 1.0f;
        return true;
    }
    case 2u: { // Dielectric
        float ri = s.front_facing ? 1.0f / m.param : m.param;
        V3 unit_direction = normalize(d);
        float cos_theta = fminf(dot(-unit_direction, s.normal), 1.0f);
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        bool cannot_refract = ri * sin_theta > 1.0f;
        if (cannot_refract || reflectance(cos_theta, ri) > rng_random(rng))
            out.direction = reflect(unit_direction, s.normal);
        else
            out.direction = refract(unit_direction, s.normal, ri);
        out.attenuation = V3{1.0f, 1.0f, 1.0f};
        out.pdf = 1.0f;
        return true;
    }
    default: // DiffuseLight
        return false;
    }
}

// Mix::scatter and ScatterResult::mix (src/material.rs:186-204,21-30; random::choose, src/random.rs:24-31):
// factor 1 -> right, factor 0 -> left, otherwise BOTH children scatter (each consuming its random draws, left
// first), then one more draw picks whose ray continues while the attenuations are blended and the pdf is
// dropped.  Children may be Mix materials themselves; DEPTH bounds the nesting (validated at scene creation).
template <int DEPTH>
__device__ __noinline__ bool scatter_material(const SceneDev &sc, int32_t mi, V3 d, const Surface &s, Rng &rng,
                                              Scatter &out) {
    const MaterialRec m = sc.materials[mi];
    if (m.kind != 4u) return scatter_leaf(sc, m, d, s, rng, out);
    if constexpr (DEPTH == 0) {
        return false; // unreachable: nesting depth is validated on the host
    } else {
        if (m.param == 1.0f) return scatter_material<DEPTH - 1>(sc, m.right, d, s, rng, out);
        if (m.param == 0.0f) return scatter_material<DEPTH - 1>(sc, m.left, d, s, rng, out);
        Scatter l, r;
        const bool got_l = scatter_material<DEPTH - 1>(sc, m.left, d, s, rng, l);
        const bool got_r = scatter_material<DEPTH - 1>(sc, m.right, d, s, rng, r);
        if (got_l && got_r) {
            const float n = rng_random(rng);
            out.direction = n < m.param ? l.direction : r.direction;
            out.attenuation = l.attenuation * m.param + r.attenuation * (1.0f - m.param);
            out.pdf = 1.0f;
            return true;
        }
        if (got_r) {
            out = r;
            return true;
        }
        if (got_l) {
            out = l;
            return true;
        }
        return false;
    }
}

} // namespace

// ------------------------------------------------------------------------------------------------------
// generate: Camera::get_ray (src/camera.rs:68-93) for every (sample, pixel) of the batch
// ------------------------------------------------------------------------------------------------------
// Which path the i-th camera ray of a batch belongs to.  Path slots are numbered sample by sample, row by row
// (slot = sample_in_batch * npix + pixel); taken in that order a wave's 64 camera rays are a 64 x 1 strip of the image.
// When both image sides are multiples of eight the rays are dealt in 8 x 8 tiles instead (P.tiles_x = width / 8): a
// wave's rays then meet the same nodes and walk for similar lengths.  A bijection of [0, n_slots) either way, so it
// changes which lane traces a path, never the path.
__device__ __forceinline__ uint32_t camera_slot(const RenderParams &P, uint32_t i) {
    if (P.tiles_x == 0u) return i;
    const uint32_t s = i / P.npix, f = i - s * P.npix;
    const uint32_t t = f >> 6, w = f & 63u;
    const uint32_t ty = t / P.tiles_x, tx = t - ty * P.tiles_x;
    return s * P.npix + (ty * 8u + (w >> 3)) * P.width + tx * 8u + (w & 7u);
}

// Camera::get_ray (src/camera.rs:68-93) for path `slot` of the batch; returns the draw counter after it.
__device__ __forceinline__ uint32_t generate_ray(const RenderParams &P, uint32_t slot, V3 &origin, V3 &dir,
                                                 uint32_t &s_local, uint32_t &pixel) {
    s_local = slot / P.npix;
    pixel = slot - s_local * P.npix;
    uint32_t j = pixel / P.width;
    uint32_t i = pixel - j * P.width;
    Rng rng = rng_for_path(P.seed_mix, pixel, P.sample_base + s_local, 0u);
    uint32_t a, b;
    rng_next_2u32(rng, a, b);
    float ox = u32_to_unit(a) - 0.5f;
    float oy = u32_to_unit(b) - 0.5f;
    V3 pixel_sample = (ld3(P.pixel_00_loc) + ld3(P.pixel_delta_u) * ((float)i + ox)) +
                      ld3(P.pixel_delta_v) * ((float)j + oy);
    if (P.defocus_angle <= 0.0f) {
        origin = ld3(P.center);
    } else {
        float px, py; // gen_unit_disk (src/vec3.rs:74-81)
        for (;;) {
            px = rng_random_range(rng, -1.0f, 1.0f);
            py = rng_random_range(rng, -1.0f, 1.0f);
            V3 p = V3{px, py, 0.0f};
            if (length_squared(p) < 1.0f) break;
        }
        origin = (ld3(P.center) + ld3(P.defocus_disk_u) * px) + ld3(P.defocus_disk_v) * py;
    }
    dir = pixel_sample - origin;
    return rng.k;
}

// Path state while a path is in flight (slot = sample_in_batch * npix + pixel):
//   ray_o = origin.xyz    | k (bits 0..27: index of the next random draw) | NaN flags (bits 28..30)
//   ray_d = direction.xyz | depth (bits 0..15) | sample_in_batch (bits 16..31)
//   beta  = throughput.xyz
// The radiance is not part of it: until a path ends its radiance is, per channel, either +0 or NaN (only
// emitters add anything else and they end the path, src/material.rs:129-137; `l += beta * 0` turns a channel
// NaN once beta is not finite, src/renderer.rs:48), so three flag bits carry it and radiance[slot] is written
// exactly once, when the path ends.
__global__ __launch_bounds__(kBlock) void k_generate(RenderParams P, PathState st, uint32_t n_slots) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_slots; i += gridDim.x * kBlock) {
        const uint32_t slot = camera_slot(P, i);
        V3 origin, dir;
        uint32_t s_local, pixel;
        uint32_t k = generate_ray(P, slot, origin, dir, s_local, pixel);
        st.ray_o[slot] = mk4(origin, u2f(k & kDrawMask));
        st.ray_d[slot] = mk4(dir, u2f(s_local << 16));
        st.beta[slot] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
    }
}

// ------------------------------------------------------------------------------------------------------
// Queues.  A queue is segmented by workgroup: block b owns entries [b * cap, b * cap + counts[b]).  A block
// appends the survivors of its own items to its own segment of the output queue, so compaction needs no
// global atomic (one returning atomic per wave on a single word serialises the whole chip) and the same
// grid must be used for every launch of a batch.  `items == nullptr` is the identity queue over [0, n): its
// 256-item chunks are dealt to the blocks round-robin, which keeps the first bounce fully coalesced and gives
// every block an even sample of the image.
// ------------------------------------------------------------------------------------------------------
struct QueueCursor {
    uint32_t off, limit, stride, seg;
    bool ident;
};

__device__ __forceinline__ QueueCursor queue_begin(const QueueView &q) {
    QueueCursor c;
    c.ident = q.items == nullptr;
    c.limit = c.ident ? q.n : q.counts[blockIdx.x];
    c.stride = c.ident ? gridDim.x * kBlock : kBlock;
    c.off = c.ident ? blockIdx.x * kBlock : 0u;
    c.seg = blockIdx.x * q.cap;
    return c;
}

// Copies the node and primitive tables into LDS (dynamic shared memory) and repoints the table pointers.
__device__ __forceinline__ void stage_scene_in_lds(const SceneDev &sc, float4 *lds, const float4 *&nodes,
                                                   const float4 *&prims) {
    const uint32_t n4_nodes = 2u * sc.n_nodes, n4_prims = 3u * sc.n_prims;
    for (uint32_t i = threadIdx.x; i < sc.n_nodes; i += blockDim.x) {
        float4 second = nodes[2 * i + 1];
        second.z = u2f(f2u(second.z) * 32u); // skip as a byte offset
        lds[2 * i] = nodes[2 * i];
        lds[2 * i + 1] = second;
    }
    for (uint32_t i = threadIdx.x; i < n4_prims; i += blockDim.x) lds[n4_nodes + i] = prims[i];
    nodes = lds;
    prims = lds + n4_nodes;
}

// The same for traverse_sorted: nodes as 48-byte records (an axis as low, high, low; skip as the absolute LDS address of
// its target), primitives behind them.  Returns the LDS address of the copy.
__device__ __forceinline__ uint32_t stage_scene_sorted(const SceneDev &sc, float4 *lds, const float4 *nodes, const float4 *&prims) {
    const uint32_t n4_nodes = 3u * sc.n_nodes, n4_prims = 3u * sc.n_prims;
    const uint32_t base = lds_address(lds);
    for (uint32_t i = threadIdx.x; i < sc.n_nodes; i += blockDim.x) {
        const float4 a = nodes[2 * i], b = nodes[2 * i + 1];
        lds[3 * i] = make_float4(a.x, a.y, a.x, a.z);
        lds[3 * i + 1] = make_float4(a.w, a.z, b.x, b.y);
        lds[3 * i + 2] = make_float4(b.x, u2f(base + f2u(b.z) * kSortedNodeBytes), b.w, 0.0f);
    }
    for (uint32_t i = threadIdx.x; i < n4_prims; i += blockDim.x) lds[n4_nodes + i] = prims[i];
    prims = lds + n4_nodes;
    return base;
}

// The same for traverse_octants: eight copies of every node (copy c: x reversed when c & 1, y when c & 2, z when c & 4),
// skip links as absolute LDS addresses of the same copy of the target.  Returns the LDS address of the copy.
__device__ __forceinline__ uint32_t stage_scene_octants(const SceneDev &sc, float4 *lds, const float4 *nodes, const float4 *&prims) {
    const uint32_t n4_nodes = 16u * sc.n_nodes, n4_prims = 3u * sc.n_prims;
    const uint32_t base = lds_address(lds);
    for (uint32_t i = threadIdx.x; i < 8u * sc.n_nodes; i += blockDim.x) {
        const uint32_t node = i >> 3, c = i & 7u;
        const float4 a = nodes[2 * node], b = nodes[2 * node + 1];
        lds[2 * i] = make_float4((c & 1u) ? a.y : a.x, (c & 1u) ? a.x : a.y, (c & 2u) ? a.w : a.z, (c & 2u) ? a.z : a.w);
        lds[2 * i + 1] = make_float4((c & 4u) ? b.y : b.x, (c & 4u) ? b.x : b.y, u2f(base + f2u(b.z) * kOctantNodeBytes + c * 32u), b.w);
    }
    for (uint32_t i = threadIdx.x; i < n4_prims; i += blockDim.x) lds[n4_nodes + i] = prims[i];
    prims = lds + n4_nodes;
    return base;
}

// ------------------------------------------------------------------------------------------------------
// extend: closest hit of every queued ray.
//
// Traversal lengths are very uneven (Cornell: 17 box tests per ray on average, but some ray in almost every
// wave of incoherent rays walks 60+ of the 71 nodes), so a wave that takes 64 rays and walks until the
// slowest is done keeps a quarter of its lanes busy.  Here every lane owns a slot of a small state machine
// instead: when kRefillLanes lanes of a wave have finished their ray, they claim the next unclaimed items of
// the block's queue segment (one LDS atomic per refill), load those rays and join the walk again.
// ------------------------------------------------------------------------------------------------------
// k_paths_lanes: rounds of walking between two service checks, per traversal
#ifndef RT_ROUNDS_CHAIN
#define RT_ROUNDS_CHAIN 16
#endif
#ifndef RT_ROUNDS_WALK
#define RT_ROUNDS_WALK 32
#endif
#ifndef RT_REFILL_LANES
#define RT_REFILL_LANES 16
#endif

template <bool COUNT, bool LDS_SCENE>
__global__ __launch_bounds__(kBlock) void k_extend(SceneDev sc, const float4 *__restrict__ ray_o,
                                                   const float4 *__restrict__ ray_d, float4 *__restrict__ hits,
                                                   QueueView q, float tmin, float tmax, ExtendCounters *counters) {
    extern __shared__ float4 s_scene[];
    __shared__ uint32_t s_next;
    __shared__ unsigned int s_cnt[3];
    const float4 *nodes = reinterpret_cast<const float4 *>(sc.nodes);
    const float4 *prims = reinterpret_cast<const float4 *>(sc.prims);
    if (LDS_SCENE) stage_scene_in_lds(sc, s_scene, nodes, prims);
    if (threadIdx.x == 0) s_next = 0u;
    if (COUNT && threadIdx.x < 3) s_cnt[threadIdx.x] = 0u;
    __syncthreads();

    // the block's pool of items: its queue segment, or its round-robin share of the identity queue's chunks
    const bool ident = q.items == nullptr;
    const uint32_t seg = blockIdx.x * q.cap;
    uint32_t pool;
    if (ident) {
        const uint32_t chunks = (q.n + kBlock - 1) / kBlock;
        pool = chunks > blockIdx.x ? ((chunks - blockIdx.x + gridDim.x - 1) / gridDim.x) * kBlock : 0u;
    } else {
        pool = q.counts[blockIdx.x];
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t end = sc.n_nodes;
    uint32_t c_nodes = 0, c_tri = 0, c_sph = 0;

    bool have = false, exhausted = false;
    uint32_t slot = 0, idx = 0;
    V3 o{}, d{};
    Slabs slabs{};
    Closest best{tmax, kMissSlot, 0.0f, 0.0f};
    for (;;) {
        const unsigned long long idle_mask = ballot(!have);
        const uint32_t n_idle = __popcll(idle_mask);
        if (!exhausted && (n_idle >= RT_REFILL_LANES)) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_next, n_idle);
            base = __shfl(base, 0);
            if (base + n_idle >= pool) exhausted = true;
            if (!have) {
                const uint32_t item = base + __popcll(idle_mask & ((1ull << lane) - 1ull));
                if (item < pool) {
                    bool ok = true;
                    if (ident) {
                        slot = ((item / kBlock) * gridDim.x + blockIdx.x) * kBlock + (item % kBlock);
                        ok = slot < q.n;
                    } else {
                        slot = q.items[seg + item];
                    }
                    if (ok) {
                        o = xyz(ray_o[slot]);
                        d = xyz(ray_d[slot]);
                        slabs = make_slabs(o, d);
                        best = Closest{tmax, kMissSlot, 0.0f, 0.0f};
                        idx = 0;
                        have = true;
                    }
                }
            }
        }
        if (ballot(have) == 0ull) {
            if (exhausted) break;
            continue;
        }
        if (have) {
            // one step of BvhNode::hit's pre-order walk (see traverse())
            float4 nlo = nodes[2 * idx];
            float4 nhi = nodes[2 * idx + 1];
            if (LDS_SCENE) nhi.z = u2f(f2u(nhi.z) >> 5); // the LDS copy stores skip links as byte offsets
            if (COUNT) ++c_nodes;
            bool hit = box_hit(nlo, nhi, slabs, tmin, best.t);
            uint32_t leaf = f2u(nhi.w);
            if (hit && leaf != kInterior) {
                uint32_t ps = leaf & ~kSphereBit;
                if (leaf & kSphereBit) {
                    if (COUNT) ++c_sph;
                    sphere_test(prims + 3 * (size_t)ps, ps, o, d, tmin, best);
                } else {
                    if (COUNT) ++c_tri;
                    tri_test(prims + 3 * (size_t)ps, ps, o, d, tmin, best);
                }
            }
            idx = hit ? idx + 1 : f2u(nhi.z);
            if (idx >= end) {
                hits[slot] = make_float4(best.t, u2f(best.slot), best.u, best.v);
                have = false;
            }
        }
    }
    if (COUNT) { // one atomic per block and counter, skipped when the block saw no work
        for (int sh = 32; sh > 0; sh >>= 1) {
            c_nodes += __shfl_down(c_nodes, sh);
            c_tri += __shfl_down(c_tri, sh);
            c_sph += __shfl_down(c_sph, sh);
        }
        if (lane == 0u) {
            if (c_nodes) atomicAdd(&s_cnt[0], c_nodes);
            if (c_tri) atomicAdd(&s_cnt[1], c_tri);
            if (c_sph) atomicAdd(&s_cnt[2], c_sph);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (s_cnt[0]) atomicAdd(&counters->node_visits, (unsigned long long)s_cnt[0]);
            if (s_cnt[1]) atomicAdd(&counters->triangle_tests, (unsigned long long)s_cnt[1]);
            if (s_cnt[2]) atomicAdd(&counters->sphere_tests, (unsigned long long)s_cnt[2]);
        }
    }
}

// World::hit for every queued ray (tests only; no BVH).
__global__ __launch_bounds__(kBlock) void k_extend_world(SceneDev sc, const float4 *__restrict__ ray_o,
                                                         const float4 *__restrict__ ray_d, float4 *__restrict__ hits,
                                                         QueueView q, float tmin, float tmax) {
    QueueCursor c = queue_begin(q);
    for (uint32_t off = c.off; off < c.limit; off += c.stride) {
        const uint32_t i = off + threadIdx.x;
        if (i >= c.limit) break;
        uint32_t slot = c.ident ? i : q.items[c.seg + i];
        Closest best = scan_world(sc, xyz(ray_o[slot]), xyz(ray_d[slot]), tmin, tmax);
        hits[slot] = make_float4(best.t, u2f(best.slot), best.u, best.v);
    }
}

// Closest hits by the walks of the fused render kernels over their LDS copies (traverse_sorted: SORTED = true,
// traverse_octants: false), one ray per lane, with the box and primitive tests counted: rt_trace_closest runs it when asked
// to (RT_TRACE_TRAVERSAL), so that those walks can be driven with chosen rays and their box decisions compared with the
// oracle's visit counts.
template <bool SORTED>
__global__ __launch_bounds__(kBlock) void k_extend_plain(SceneDev sc, const float4 *__restrict__ ray_o, const float4 *__restrict__ ray_d,
                                                         float4 *__restrict__ hits, uint32_t n, float tmin, float tmax,
                                                         ExtendCounters *counters) {
    extern __shared__ float4 s_scene[];
    const float4 *nodes = reinterpret_cast<const float4 *>(sc.nodes);
    const float4 *prims = reinterpret_cast<const float4 *>(sc.prims);
    uint32_t sorted_nodes = 0, octant_nodes = 0;
    if (SORTED)
        sorted_nodes = stage_scene_sorted(sc, s_scene, nodes, prims);
    else
        octant_nodes = stage_scene_octants(sc, s_scene, nodes, prims);
    __syncthreads();
    uint32_t visits[3] = {0u, 0u, 0u};
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const V3 o = xyz(ray_o[i]), d = xyz(ray_d[i]);
        const Closest best = SORTED ? traverse_sorted<true>(sorted_nodes, prims, sc.n_nodes, o, d, tmin, tmax, visits)
                                    : traverse_octants<true>(octant_nodes, prims, sc.n_nodes, o, d, tmin, tmax, visits);
        hits[i] = make_float4(best.t, u2f(best.slot), best.u, best.v);
    }
    if (counters) {
        if (visits[0]) atomicAdd(&counters->node_visits, (unsigned long long)visits[0]);
        if (visits[1]) atomicAdd(&counters->triangle_tests, (unsigned long long)visits[1]);
        if (visits[2]) atomicAdd(&counters->sphere_tests, (unsigned long long)visits[2]);
    }
}

// ------------------------------------------------------------------------------------------------------
// shading: the body of Renderer::ray_color's loop after `world.hit` (src/renderer.rs:46-66) for one path.
// Writes the path's next state (and returns true) or its final radiance (and returns false).
// ------------------------------------------------------------------------------------------------------
//
// Early miss: when the scattered ray does not even enter the scene's root box (a fifth of Cornell's diffuse
// bounces leave the box at once), the next `world.hit` is known to return None: the path would come back next
// bounce only to be assigned the background colour.  That assignment is done here instead and the function
// returns kPathEndedByEarlyMiss so the caller can still count the ray; the sums are unchanged.
enum : int { kPathEnded = 0, kPathAlive = 1, kPathEndedByEarlyMiss = 2 };

// The segment of a path that is being traced: its ray, throughput, the index of its next random draw, the NaN
// flags of its radiance so far and its depth.
struct Segment {
    V3 o, d, beta;
    uint32_t k, flags, depth;
};

// Shades the hit `best` of segment `g`.  Returns kPathAlive with `g` replaced by the path's next segment, or one
// of the two ended outcomes with the path's final radiance in `l`.
template <bool NEED_UV, bool HAS_MIX>
__device__ __forceinline__ int shade_segment(const SceneDev &sc, float4 root_lo, float4 root_hi,
                                             const float4 *__restrict__ prims, const RenderParams &P, Segment &g,
                                             uint32_t s_local, uint32_t pixel, const Closest &best, V3 &l) {
    bool alive = false;
    bool early_miss = false;
    const float qnan = __builtin_nanf("");
    l = V3{(g.flags & 1u) ? qnan : 0.0f, (g.flags & 2u) ? qnan : 0.0f, (g.flags & 4u) ? qnan : 0.0f};
    if (best.slot == kMissSlot) {
        l = ld3(P.background); // src/renderer.rs:59-63: the background REPLACES the radiance gathered so far
    } else {
        Surface s = make_surface<NEED_UV>(sc, prims, g.o, g.d, best.t, best.slot, best.u, best.v);
        const MaterialRec m = sc.materials[s.material];
        // src/renderer.rs:48: l += beta * emit.  Non-emitters contribute beta * 0, which is NaN once beta holds
        // an infinity, exactly as in the reference.
        V3 emit = V3{0.0f, 0.0f, 0.0f};
        if (m.kind == 3u) emit = material_color(sc, m, s) * m.param;
        l = l + g.beta * emit;
        Rng rng = rng_for_path(P.seed_mix, pixel, P.sample_base + s_local, g.k);
        Scatter sr;
        const bool scattered = HAS_MIX ? scatter_material<4>(sc, (int32_t)s.material, g.d, s, rng, sr)
                                       : scatter_leaf(sc, m, g.d, s, rng, sr);
        if (scattered) {
            V3 f = (sr.attenuation * fabsf(dot(sr.direction, s.normal))) / sr.pdf; // src/renderer.rs:51-54
            V3 beta = g.beta * f;
            const uint32_t depth = g.depth + 1u;
            // src/renderer.rs:39-41: `while beta != 0` then `depth >= max_depth`
            alive = !(beta.x == 0.0f && beta.y == 0.0f && beta.z == 0.0f) && depth < P.max_depth;
            if (alive) {
                if (!box_hit_once(root_lo, root_hi, s.point, sr.direction, P.z_near, P.z_far)) {
                    alive = false;
                    early_miss = true;
                    l = ld3(P.background);
                }
            }
            if (alive) {
                // a scattering material emits nothing, so l is still +0 or NaN per channel here
                g.flags = (l.x != l.x ? 1u : 0u) | (l.y != l.y ? 2u : 0u) | (l.z != l.z ? 4u : 0u);
                g.o = s.point;
                g.d = sr.direction;
                g.beta = beta;
                g.k = rng.k;
                g.depth = depth;
            }
        }
    }
    return alive ? kPathAlive : (early_miss ? kPathEndedByEarlyMiss : kPathEnded);
}

// The same with the path's state kept in device memory: stores the next segment, or the final radiance.
template <bool NEED_UV, bool HAS_MIX>
__device__ __forceinline__ int shade_path(const SceneDev &sc, float4 root_lo, float4 root_hi,
                                          const float4 *__restrict__ prims, const RenderParams &P, const PathState &st,
                                          uint32_t slot, V3 o, V3 d, V3 beta, uint32_t k, uint32_t flags, uint32_t depth,
                                          uint32_t s_local, uint32_t pixel, const Closest &best) {
    Segment g{o, d, beta, k, flags, depth};
    V3 l;
    const int outcome = shade_segment<NEED_UV, HAS_MIX>(sc, root_lo, root_hi, prims, P, g, s_local, pixel, best, l);
    if (outcome == kPathAlive) {
        st.ray_o[slot] = mk4(g.o, u2f((g.k & kDrawMask) | (g.flags << kDrawBits)));
        st.ray_d[slot] = mk4(g.d, u2f((s_local << 16) | g.depth));
        st.beta[slot] = mk4(g.beta, 0.0f);
    } else {
        st.radiance[slot] = mk4(l, 0.0f);
    }
    return outcome;
}

// Appends the surviving lanes' slots to the block's output segment: ranks from ballot + popcount, one LDS
// atomic per wave.  Must be reached by all 64 lanes of the wave.
__device__ __forceinline__ void compact_survivors(int outcome, uint32_t slot, uint32_t lane, uint32_t *s_out,
                                                  uint32_t *s_early, uint32_t *__restrict__ queue_out, uint32_t seg) {
    const bool alive = outcome == kPathAlive;
    unsigned long long early = ballot(outcome == kPathEndedByEarlyMiss);
    if (early && lane == 0) atomicAdd(s_early, (uint32_t)__popcll(early));
    unsigned long long mask = ballot(alive);
    if (mask) {
        uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
        uint32_t total = __popcll(mask);
        uint32_t start = 0;
        if (lane == 0) start = atomicAdd(s_out, total);
        start = __shfl(start, 0);
        if (alive) queue_out[seg + start + rank] = slot;
    }
}

// shade: every queued path whose hit record extend has written.
template <bool NEED_UV, bool HAS_MIX>
__global__ __launch_bounds__(kBlock) void k_shade(SceneDev sc, RenderParams P, PathState st, QueueView q,
                                                  uint32_t *__restrict__ queue_out, uint32_t *__restrict__ counts_out,
                                                  uint32_t *__restrict__ early_out) {
    __shared__ uint32_t s_out, s_early;
    if (threadIdx.x == 0) {
        s_out = 0u;
        s_early = 0u;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const float4 *nodes = reinterpret_cast<const float4 *>(sc.nodes);
    const float4 *prims = reinterpret_cast<const float4 *>(sc.prims);
    QueueCursor c = queue_begin(q);
    // Whole waves iterate together so that the ballot in compact_survivors always sees all 64 lanes.
    for (uint32_t off = c.off; off < c.limit; off += c.stride) {
        const uint32_t i = off + threadIdx.x;
        int alive = kPathEnded;
        uint32_t slot = 0;
        if (i < c.limit) {
            slot = c.ident ? i : q.items[c.seg + i];
            float4 h = st.hit[slot];
            float4 o4 = st.ray_o[slot];
            float4 d4 = st.ray_d[slot];
            float4 b4 = st.beta[slot];
            Closest best{h.x, f2u(h.y), h.z, h.w};
            const uint32_t s_local = f2u(d4.w) >> 16;
            alive = shade_path<NEED_UV, HAS_MIX>(sc, nodes[0], nodes[1], prims, P, st, slot, xyz(o4), xyz(d4), xyz(b4), f2u(o4.w) & kDrawMask,
                                        f2u(o4.w) >> kDrawBits, f2u(d4.w) & 0xFFFFu, s_local, slot - s_local * P.npix, best);
        }
        compact_survivors(alive, slot, lane, &s_out, &s_early, queue_out, c.seg);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        counts_out[blockIdx.x] = s_out;
        early_out[blockIdx.x] = s_early;
    }
}

// ------------------------------------------------------------------------------------------------------
// bounce: generate (first bounce only) + extend + shade of one path segment in a single kernel: the ray and
// the hit stay in registers between traversal and shading.  With LDS_SCENE the node and primitive tables are
// first copied into LDS and traversal reads them there.
// ------------------------------------------------------------------------------------------------------
// Six waves per SIMD: the kernel needs 81 VGPRs unconstrained, one over the 80 that let a sixth wave in.
// STACK: 0 = the stackless walk over the 32-byte node table; K > 0 = the stack traversal over the 128-byte records
// with K stack entries per lane in LDS (tables in device memory only); -1 = the stackless walk over the walk table,
// whose treetop (sc.top_bytes) the block stages in LDS.
template <bool FIRST, bool LDS_SCENE, bool NEED_UV, bool HAS_MIX = false, int STACK = 0>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(HAS_MIX ? 4 : 6))) void k_bounce(SceneDev sc, RenderParams P, PathState st, QueueView q,
                                                   uint32_t *__restrict__ queue_out, uint32_t *__restrict__ counts_out,
                                                   uint32_t *__restrict__ early_out, StackSpill spill) {
    static_assert(!(LDS_SCENE && STACK), "the stack traversal and the walk table read their tables from device memory");
    extern __shared__ float4 s_scene[];
    __shared__ uint32_t s_out, s_early;
    const float4 *nodes = reinterpret_cast<const float4 *>(sc.nodes);
    const float4 *prims = reinterpret_cast<const float4 *>(sc.prims);
    const float4 root_lo = nodes[0], root_hi = nodes[1]; // the root box in the table's own layout, for early misses
    uint32_t sorted_nodes = 0;
    if (LDS_SCENE) sorted_nodes = stage_scene_sorted(sc, s_scene, nodes, prims);
    const float4 *__restrict__ walk_nodes = nullptr;
    if constexpr (STACK < 0) {
        walk_nodes = uniform_pointer(reinterpret_cast<const float4 *>(sc.walk_nodes));
        const uint32_t n4 = sc.top_bytes / 16u;
        for (uint32_t i = threadIdx.x; i < n4; i += kBlock) s_scene[i] = walk_nodes[i];
    }
    if (threadIdx.x == 0) {
        s_out = 0u;
        s_early = 0u;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    QueueCursor c = queue_begin(q);
    for (uint32_t off = c.off; off < c.limit; off += c.stride) {
        const uint32_t i = off + threadIdx.x;
        int alive = kPathEnded;
        uint32_t slot = 0;
        if (i < c.limit) {
            V3 o, d, beta;
            uint32_t k, flags, depth, s_local, pixel;
            if (FIRST) {
                slot = camera_slot(P, i);
                k = generate_ray(P, slot, o, d, s_local, pixel);
                beta = V3{1.0f, 1.0f, 1.0f};
                flags = 0u;
                depth = 0u;
            } else {
                slot = q.items[c.seg + i];
                float4 o4 = st.ray_o[slot], d4 = st.ray_d[slot];
                o = xyz(o4);
                d = xyz(d4);
                beta = xyz(st.beta[slot]);
                k = f2u(o4.w) & kDrawMask;
                flags = f2u(o4.w) >> kDrawBits;
                depth = f2u(d4.w) & 0xFFFFu;
                s_local = f2u(d4.w) >> 16;
                pixel = slot - s_local * P.npix;
            }
            Closest best;
            if constexpr (STACK > 0) {
                best = traverse_stack<(STACK > 0 ? STACK : 1), kBlock>(reinterpret_cast<const float4 *>(sc.chain), prims, root_lo, root_hi, sc.root_ref,
                                                                      o, d, P.z_near, P.z_far, reinterpret_cast<uint2 *>(s_scene), spill);
            } else if constexpr (STACK < 0) {
                best = traverse_walk(walk_nodes, lds_address(s_scene), sc.top_bytes, prims, o, d, P.z_near, P.z_far);
            } else {
                if constexpr (LDS_SCENE) {
                    best = traverse_sorted(sorted_nodes, prims, sc.n_nodes, o, d, P.z_near, P.z_far);
                } else {
                    best = traverse(nodes, prims, sc.n_nodes, o, d, P.z_near, P.z_far);
                }
            }
            alive = shade_path<NEED_UV, HAS_MIX>(sc, root_lo, root_hi, prims, P, st, slot, o, d, beta, k, flags, depth, s_local, pixel, best);
        }
        compact_survivors(alive, slot, lane, &s_out, &s_early, queue_out, c.seg);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        counts_out[blockIdx.x] = s_out;
        early_out[blockIdx.x] = s_early;
    }
}

// ------------------------------------------------------------------------------------------------------
// paths: every bounce of a batch in ONE launch.  A block only ever consumes the queue segment it produced
// itself (see QueueView), so the bounce loop needs no grid-wide synchronisation: the block generates its share
// of camera rays, then keeps bouncing its own survivors between its segments of the two queues until none
// are left or max_depth is reached.  `counts` / `early` are the [max_depth + 1][gridDim.x] tables the
// per-bounce kernels fill; rows a block never reaches must have been zeroed by the host.
// ------------------------------------------------------------------------------------------------------
// THREADS: 256, or 1024 for scenes whose tables take a large part of a CU's LDS (one copy then serves sixteen
// waves instead of four); work is claimed per wave, so nothing else depends on the block size.
// WAVES: waves per SIMD the register budget is set for.  Six fit without spilling and suit scenes served by LDS
// or the caches; scenes read from HBM gain more from eight waves in flight than the few spilled registers cost.
// HAS_MIX: the scene holds Mix materials (evaluated by scatter_material instead of scatter_leaf).
// OCTANTS: the LDS copy holds eight copies of every node (traverse_octants; scenes of a few dozen nodes).
template <bool LDS_SCENE, bool NEED_UV, int THREADS = kBlock, int WAVES = 6, bool HAS_MIX = false, bool OCTANTS = false>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu((THREADS == kBlock && !HAS_MIX) ? WAVES : 4))) void k_paths(
    SceneDev sc, RenderParams P, PathState st, uint32_t n_slots, uint32_t cap, uint32_t *__restrict__ queue0,
    uint32_t *__restrict__ queue1, uint32_t *__restrict__ counts, uint32_t *__restrict__ early) {
    extern __shared__ float4 s_scene[];
    __shared__ uint32_t s_out, s_early, s_next;
    const float4 *nodes = reinterpret_cast<const float4 *>(sc.nodes);
    const float4 *prims = reinterpret_cast<const float4 *>(sc.prims);
    const float4 root_lo = nodes[0], root_hi = nodes[1];
    uint32_t sorted_nodes = 0, octant_nodes = 0;
    if constexpr (LDS_SCENE) {
        if constexpr (OCTANTS)
            octant_nodes = stage_scene_octants(sc, s_scene, nodes, prims);
        else
            sorted_nodes = stage_scene_sorted(sc, s_scene, nodes, prims);
    }
    auto closest_hit = [&](V3 o, V3 d) {
        if constexpr (LDS_SCENE && OCTANTS) {
            return traverse_octants(octant_nodes, prims, sc.n_nodes, o, d, P.z_near, P.z_far);
        } else if constexpr (LDS_SCENE) {
            return traverse_sorted(sorted_nodes, prims, sc.n_nodes, o, d, P.z_near, P.z_far);
        } else {
            return traverse(nodes, prims, sc.n_nodes, o, d, P.z_near, P.z_far);
        }
    };
    if (threadIdx.x == 0) {
        s_out = 0u;
        s_early = 0u;
        s_next = 0u;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t seg = blockIdx.x * cap;

    // Within a bounce the block's waves claim 64 items at a time from a shared cursor, so a wave that drew
    // long traversals does not leave the others waiting at the end of the bounce.
    //
    // bounce 0: the block's round-robin share of the 256-slot chunks of the batch, in quarter chunks
    const uint32_t chunks = (n_slots + kBlock - 1) / kBlock;
    const uint32_t my_quarters = chunks > blockIdx.x ? ((chunks - blockIdx.x + gridDim.x - 1) / gridDim.x) * 4u : 0u;
    for (;;) {
        uint32_t quarter = 0;
        if (lane == 0) quarter = atomicAdd(&s_next, 1u);
        quarter = __shfl(quarter, 0);
        if (quarter >= my_quarters) break;
        const uint32_t item = ((quarter >> 2) * gridDim.x + blockIdx.x) * kBlock + (quarter & 3u) * 64u + lane;
        const uint32_t slot = item < n_slots ? camera_slot(P, item) : item;
        int alive = kPathEnded;
        if (item < n_slots) {
            V3 o, d;
            uint32_t s_local, pixel;
            const uint32_t k = generate_ray(P, slot, o, d, s_local, pixel);
            Closest best = closest_hit(o, d);
            alive = shade_path<NEED_UV, HAS_MIX>(sc, root_lo, root_hi, prims, P, st, slot, o, d, V3{1.0f, 1.0f, 1.0f}, k, 0u, 0u,
                                                 s_local, pixel, best);
        }
        compact_survivors(alive, slot, lane, &s_out, &s_early, queue0, seg);
    }
    for (uint32_t it = 1;; ++it) {
        __syncthreads(); // the bounce's appends, and the path state behind them, are complete and visible
        const uint32_t n_in = s_out;
        if (threadIdx.x == 0) {
            counts[(size_t)it * gridDim.x + blockIdx.x] = n_in;
            early[(size_t)it * gridDim.x + blockIdx.x] = s_early;
        }
        if (n_in == 0u || it >= P.max_depth) break;
        __syncthreads(); // everyone has read the count before it is reset
        if (threadIdx.x == 0) {
            s_out = 0u;
            s_early = 0u;
            s_next = 0u;
        }
        __syncthreads();
        const uint32_t *__restrict__ q_in = (it & 1u) ? queue0 : queue1;
        uint32_t *__restrict__ q_out = (it & 1u) ? queue1 : queue0;
        for (;;) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_next, 64u);
            base = __shfl(base, 0);
            if (base >= n_in) break;
            const uint32_t i = base + lane;
            int alive = kPathEnded;
            uint32_t slot = 0;
            if (i < n_in) {
                slot = q_in[seg + i];
                const float4 o4 = st.ray_o[slot], d4 = st.ray_d[slot];
                const V3 o = xyz(o4), d = xyz(d4);
                const uint32_t s_local = f2u(d4.w) >> 16;
                Closest best = closest_hit(o, d);
                alive = shade_path<NEED_UV, HAS_MIX>(sc, root_lo, root_hi, prims, P, st, slot, o, d, xyz(st.beta[slot]),
                                                     f2u(o4.w) & kDrawMask, f2u(o4.w) >> kDrawBits, f2u(d4.w) & 0xFFFFu, s_local,
                                                     slot - s_local * P.npix, best);
            }
            compact_survivors(alive, slot, lane, &s_out, &s_early, q_out, seg);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// paths, one lane per path: for scenes read from HBM.  There a ray's walk is a chain of dependent fetches whose
// length varies several-fold from ray to ray, so a wave that starts 64 rays together and waits for the longest
// keeps a sixth of its lanes busy.  Here a lane owns a PATH and keeps it in registers from the camera ray to its
// end: when enough of a wave's lanes have finished their walk they are shaded where they are and start their
// next segment at once; lanes whose path has ended take the next path of the block's share of the batch.  No
// queue, no path state in device memory: only the final radiance is written.  Samples are still added per pixel
// in sample order by k_accumulate, so the sums do not depend on which lane traced what.
// `ray_counts[b]`: rays beyond the first of each path that block b traced (settled early misses included).
// ------------------------------------------------------------------------------------------------------
// FROM_CAMERA: lanes take paths from the camera ray on (the block's round-robin share of the batch).  Otherwise
// bounce 0 has been traced by k_bounce (64 neighbouring pixels per wave in lockstep: their walks touch the same
// nodes, which the vector cache then fetches once per wave) and lanes take the surviving paths from the block's
// segment of `queue` (`pool[b]` entries), whose first rays the caller has counted already.
// CHAIN: the stack traversal over the 128-byte chain records (scenes read from HBM); otherwise the stackless walk
// over the 32-byte node table (scenes that sit in the L2s).
// THREADS: threads per block.  The walk table's treetop is staged once per block, so larger blocks share a larger
// treetop; work is claimed per wave, nothing else depends on the block size.
// `segs` (!FROM_CAMERA): the block takes the survivors of `segs` consecutive blocks of the bounce-0 launch, i.e. the
// queue segments [blockIdx.x * segs, (blockIdx.x + 1) * segs) (at most kMaxSegs).
constexpr uint32_t kMaxSegs = 32;
template <bool NEED_UV, bool HAS_MIX, int WAVES, int K, bool FROM_CAMERA, bool CHAIN = true, int THREADS = kBlock>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(HAS_MIX ? 4 : WAVES))) void k_paths_lanes(
    SceneDev sc, RenderParams P, PathState st, uint32_t n_slots, uint32_t cap, const uint32_t *__restrict__ queue,
    const uint32_t *__restrict__ pool, uint32_t *__restrict__ ray_counts, StackSpill spill, uint32_t segs) {
    extern __shared__ float4 s_scene[];
    __shared__ uint32_t s_next, s_rays;
    __shared__ uint32_t s_first[kMaxSegs + 1]; // !FROM_CAMERA: item index of the first entry of each of the block's segments
    const float4 *__restrict__ chain = uniform_pointer(reinterpret_cast<const float4 *>(CHAIN ? static_cast<const void *>(sc.chain) : static_cast<const void *>(sc.walk_nodes)));
    const float4 *prims = reinterpret_cast<const float4 *>(sc.prims);
    const float4 root_lo = reinterpret_cast<const float4 *>(sc.nodes)[0], root_hi = reinterpret_cast<const float4 *>(sc.nodes)[1];
    if (threadIdx.x == 0) {
        s_next = 0u;
        s_rays = 0u;
        if (!FROM_CAMERA) {
            uint32_t acc = 0;
            for (uint32_t i = 0; i < segs; ++i) {
                s_first[i] = acc;
                acc += pool[blockIdx.x * segs + i];
            }
            s_first[segs] = acc;
        }
    }
    uint32_t chain_lds = kRefNone, root_ref = sc.root_ref;
    if constexpr (!CHAIN) { // the treetop of the walk table
        const uint32_t n4 = sc.top_bytes / 16u;
        for (uint32_t i = threadIdx.x; i < n4; i += THREADS) s_scene[i] = chain[i];
        if (sc.prims_in_lds) { // a scene small enough for its primitives to follow the (whole) node table
            for (uint32_t i = threadIdx.x; i < 3u * sc.n_prims; i += THREADS) s_scene[n4 + i] = prims[i];
            prims = s_scene + n4;
        }
    } else if (sc.n_chain_top != 0u) { // the top chain records, behind the stack area
        float4 *dst = s_scene + (size_t)K * THREADS / 2;
        const float4 *__restrict__ src = reinterpret_cast<const float4 *>(sc.chain_top);
        for (uint32_t i = threadIdx.x; i < sc.n_chain_top * 8u; i += THREADS) dst[i] = src[i];
        chain_lds = lds_address(dst);
        if ((root_ref & kRefLeaf) == 0u) root_ref = kRefTop; // slot 0 is the root's record
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t my_items;
    if (FROM_CAMERA) {
        const uint32_t chunks = (n_slots + kBlock - 1) / kBlock;
        my_items = chunks > blockIdx.x ? ((chunks - blockIdx.x + gridDim.x - 1) / gridDim.x) * kBlock : 0u;
    } else {
        my_items = s_first[segs];
    }

    typename std::conditional<CHAIN, StackWalk<K, THREADS>, TopWalk>::type walk;
    walk.init(reinterpret_cast<uint2 *>(s_scene), spill);
    Segment g{};
    uint32_t slot = 0, s_local = 0, pixel = 0, n_rays = 0;

    bool have = false, exhausted = false;
    walk.cur = kRefNone;
    for (;;) {
        const bool finished = have && walk.cur == kRefNone;
        const uint32_t n_fin = __popcll(ballot(finished));
        const uint32_t n_empty = __popcll(ballot(!have));
        const uint32_t n_busy = 64u - n_fin - n_empty;
        // Lanes that could be put to work now, against the lanes this wave can still use: serve them once they are
        // a quarter of the wave, or when nothing else is left to do.
        const uint32_t idle = n_fin + (exhausted ? 0u : n_empty);
        const uint32_t usable = 64u - (exhausted ? n_empty : 0u);
        if (n_busy == 0u || idle * RT_SERVICE_DIV >= usable) {
#ifdef RT_LANE_STATS
            if (lane == 0u) {
                atomicAdd(&g_lane_stats[7], 1ull);
                atomicAdd(&g_lane_stats[8], (unsigned long long)n_fin);
            }
#endif
            if (n_fin && finished) {
                V3 l;
                g.o = walk.o;
                g.d = walk.d;
                const int outcome = shade_segment<NEED_UV, HAS_MIX>(sc, root_lo, root_hi, prims, P, g, s_local, pixel, walk.best, l);
                if (outcome == kPathAlive) {
                    walk.begin(root_lo, root_hi, root_ref, g.o, g.d, P.z_near, P.z_far);
                    ++n_rays;
                } else {
                    st.radiance[slot] = mk4(l, 0.0f);
                    have = false;
                    if (outcome == kPathEndedByEarlyMiss) ++n_rays;
                }
            }
            const unsigned long long empty_mask = ballot(!have);
            if (!exhausted && empty_mask) {
                const uint32_t want = __popcll(empty_mask);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&s_next, want);
                base = __shfl(base, 0);
                if (base + want >= my_items) exhausted = true;
                if (!have) {
                    const uint32_t item = base + __popcll(empty_mask & ((1ull << lane) - 1ull));
                    if (!FROM_CAMERA) {
                        if (item < my_items) {
                            uint32_t sg = 0;
                            while (item >= s_first[sg + 1]) ++sg;
                            slot = queue[(size_t)(blockIdx.x * segs + sg) * cap + (item - s_first[sg])];
                            const float4 o4 = st.ray_o[slot], d4 = st.ray_d[slot];
                            g.o = xyz(o4);
                            g.d = xyz(d4);
                            g.beta = xyz(st.beta[slot]);
                            g.k = f2u(o4.w) & kDrawMask;
                            g.flags = f2u(o4.w) >> kDrawBits;
                            g.depth = f2u(d4.w) & 0xFFFFu;
                            s_local = f2u(d4.w) >> 16;
                            pixel = slot - s_local * P.npix;
                            walk.begin(root_lo, root_hi, root_ref, g.o, g.d, P.z_near, P.z_far);
                            have = true;
                        }
                    } else {
                        const uint32_t at = ((item / kBlock) * gridDim.x + blockIdx.x) * kBlock + (item % kBlock);
                        if (item < my_items && at < n_slots) {
                            slot = camera_slot(P, at);
                            g.k = generate_ray(P, slot, g.o, g.d, s_local, pixel);
                            g.beta = V3{1.0f, 1.0f, 1.0f};
                            g.flags = 0u;
                            g.depth = 0u;
                            walk.begin(root_lo, root_hi, root_ref, g.o, g.d, P.z_near, P.z_far);
                            have = true;
                        }
                    }
                }
            }
            if (ballot(have) == 0ull) {
                if (exhausted) break;
                continue;
            }
        }
        // Several rounds of walking between two looks at who has finished: the look (four ballots and the branches
        // around the service code) costs about as much as a box test, and lanes served in larger groups are shaded in
        // fuller waves.  Measured with 1 / 8 / 32 / 64 / 128 rounds: teapot 3.27 / 3.66 / 3.79 / 3.79 / 3.71 Grays/s (stackless
        // walk); chain records, 8 / 16 / 32 / 64 rounds: 10 M soup 0.614 / 0.623 / 0.624 / 0.614, 1 M soup 1.14 / 1.15 / 1.12 / 1.01.
        // In a round, lanes whose next step is a primitive test wait until they are a quarter of the walking lanes (or
        // nothing else is left): a wave then runs the node code or the triangle code in a round, not both.
#pragma nounroll
        for (int r = 0; r < (CHAIN ? RT_ROUNDS_CHAIN : RT_ROUNDS_WALK); ++r) {
            // a lane without a path has cur == kRefNone, which is neither: `have` needs no look here
            const bool leaf_lane = walk.at_leaf(), node_lane = walk.at_node();
            const uint32_t n_leaf = __popcll(ballot(leaf_lane)), n_node = __popcll(ballot(node_lane));
#ifdef RT_LANE_STATS
            if (lane == 0u) {
                const bool leaf_round = n_node == 0u || n_leaf * RT_LEAF_DIV >= n_node + n_leaf;
                const uint32_t n_have = __popcll(ballot(have));
                atomicAdd(&g_lane_stats[leaf_round ? 2 : 0], 1ull);
                atomicAdd(&g_lane_stats[leaf_round ? 3 : 1], (unsigned long long)(leaf_round ? n_leaf : n_node));
                atomicAdd(&g_lane_stats[4], (unsigned long long)(n_have - n_leaf - n_node));
                atomicAdd(&g_lane_stats[5], (unsigned long long)(64u - n_have));
                if (!leaf_round) atomicAdd(&g_lane_stats[6], (unsigned long long)n_leaf);
            }
#endif
            if (n_node == 0u || n_leaf * RT_LEAF_DIV >= n_node + n_leaf) {
                if (leaf_lane) walk.step_leaf(prims);
            } else {
                if constexpr (CHAIN) {
                    if (node_lane) walk.step_node(chain, chain_lds);
                } else {
                    if (node_lane) walk.step_node(chain, lds_address(s_scene), sc.top_bytes);
                }
            }
        }
    }
    for (int sh = 32; sh > 0; sh >>= 1) n_rays += __shfl_down(n_rays, sh);
    if (lane == 0u && n_rays) atomicAdd(&s_rays, n_rays);
    __syncthreads();
    if (threadIdx.x == 0) ray_counts[blockIdx.x] = s_rays;
}

// Closest hits of a batch of rays, one lane per ray, in the loop of k_paths_lanes without the shading: for scenes
// whose tables are read from HBM (CHAIN: stack traversal over the chain records) or through the L2s (stackless walk).
// Ray i's hit goes to hits[i]; block b owns the 256-ray chunks b, b + gridDim.x, ...
template <int K, bool CHAIN, int THREADS = kBlock>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(CHAIN ? 4 : 6))) void k_extend_lanes(
    SceneDev sc, const float4 *__restrict__ ray_o, const float4 *__restrict__ ray_d, float4 *__restrict__ hits, uint32_t n,
    float tmin, float tmax, StackSpill spill) {
    extern __shared__ float4 s_scene[];
    __shared__ uint32_t s_next;
    const float4 *__restrict__ chain = uniform_pointer(reinterpret_cast<const float4 *>(CHAIN ? static_cast<const void *>(sc.chain) : static_cast<const void *>(sc.walk_nodes)));
    const float4 *__restrict__ prims = reinterpret_cast<const float4 *>(sc.prims);
    const float4 root_lo = reinterpret_cast<const float4 *>(sc.nodes)[0], root_hi = reinterpret_cast<const float4 *>(sc.nodes)[1];
    if (threadIdx.x == 0) s_next = 0u;
    uint32_t chain_lds = kRefNone, root_ref = sc.root_ref;
    if constexpr (!CHAIN) { // the treetop of the walk table
        const uint32_t n4 = sc.top_bytes / 16u;
        for (uint32_t i = threadIdx.x; i < n4; i += THREADS) s_scene[i] = chain[i];
    } else if (sc.n_chain_top != 0u) { // the top chain records, behind the stack area
        float4 *dst = s_scene + (size_t)K * THREADS / 2;
        const float4 *__restrict__ src = reinterpret_cast<const float4 *>(sc.chain_top);
        for (uint32_t i = threadIdx.x; i < sc.n_chain_top * 8u; i += THREADS) dst[i] = src[i];
        chain_lds = lds_address(dst);
        if ((root_ref & kRefLeaf) == 0u) root_ref = kRefTop; // slot 0 is the root's record
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t chunks = (n + kBlock - 1) / kBlock;
    const uint32_t pool = chunks > blockIdx.x ? ((chunks - blockIdx.x + gridDim.x - 1) / gridDim.x) * kBlock : 0u;

    typename std::conditional<CHAIN, StackWalk<K, THREADS>, TopWalk>::type walk;
    walk.init(reinterpret_cast<uint2 *>(s_scene), spill);
    uint32_t slot = 0;
    bool have = false, exhausted = false;
    walk.cur = kRefNone;
    for (;;) {
        const bool finished = have && walk.cur == kRefNone;
        const uint32_t n_fin = __popcll(ballot(finished));
        const uint32_t n_empty = __popcll(ballot(!have));
        const uint32_t n_busy = 64u - n_fin - n_empty;
        const uint32_t idle = n_fin + (exhausted ? 0u : n_empty);
        const uint32_t usable = 64u - (exhausted ? n_empty : 0u);
        if (n_busy == 0u || idle * RT_SERVICE_DIV >= usable) {
            if (finished) {
                hits[slot] = make_float4(walk.best.t, u2f(walk.best.slot), walk.best.u, walk.best.v);
                have = false;
            }
            const unsigned long long empty_mask = ballot(!have);
            if (!exhausted && empty_mask) {
                const uint32_t want = __popcll(empty_mask);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&s_next, want);
                base = __shfl(base, 0);
                if (base + want >= pool) exhausted = true;
                if (!have) {
                    const uint32_t item = base + __popcll(empty_mask & ((1ull << lane) - 1ull));
                    slot = ((item / kBlock) * gridDim.x + blockIdx.x) * kBlock + (item % kBlock);
                    if (item < pool && slot < n) {
                        walk.begin(root_lo, root_hi, root_ref, xyz(ray_o[slot]), xyz(ray_d[slot]), tmin, tmax);
                        have = true;
                    }
                }
            }
            if (ballot(have) == 0ull) {
                if (exhausted) break;
                continue;
            }
        }
#pragma nounroll
        for (int r = 0; r < (CHAIN ? RT_ROUNDS_CHAIN : RT_ROUNDS_WALK); ++r) {
            // a lane without a path has cur == kRefNone, which is neither: `have` needs no look here
            const bool leaf_lane = walk.at_leaf(), node_lane = walk.at_node();
            const uint32_t n_leaf = __popcll(ballot(leaf_lane)), n_node = __popcll(ballot(node_lane));
            if (n_node == 0u || n_leaf * RT_LEAF_DIV >= n_node + n_leaf) {
                if (leaf_lane) walk.step_leaf(prims);
            } else {
                if constexpr (CHAIN) {
                    if (node_lane) walk.step_node(chain, chain_lds);
                } else {
                    if (node_lane) walk.step_node(chain, lds_address(s_scene), sc.top_bytes);
                }
            }
        }
    }
}

// Per-bounce ray totals of a batch: totals[it] = sum over blocks of counts[it][b] + early[it][b] (queued rays
// plus rays whose miss was settled a bounce early).
__global__ __launch_bounds__(kBlock) void k_sum_counts(const uint32_t *__restrict__ counts,
                                                       const uint32_t *__restrict__ early, uint32_t n_blocks,
                                                       unsigned long long *__restrict__ totals) {
    __shared__ unsigned long long s_sum;
    if (threadIdx.x == 0) s_sum = 0ull;
    __syncthreads();
    unsigned long long acc = 0;
    for (uint32_t b = threadIdx.x; b < n_blocks; b += kBlock)
        acc += (unsigned long long)counts[(size_t)blockIdx.x * n_blocks + b] + early[(size_t)blockIdx.x * n_blocks + b];
    atomicAdd(&s_sum, acc);
    __syncthreads();
    if (threadIdx.x == 0) totals[blockIdx.x] = s_sum;
}

// ------------------------------------------------------------------------------------------------------
// accumulate: fold the batch's per-path radiance into the image, samples in ascending order
// (src/renderer.rs:87-93 for LINEAR_MEAN; :132-137 with linear_to_gamma :159-165 for GAMMA_PER_SAMPLE)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_accumulate(RenderParams P, const float4 *__restrict__ radiance,
                                                       float *__restrict__ accum) {
    for (uint32_t pixel = blockIdx.x * kBlock + threadIdx.x; pixel < P.npix; pixel += gridDim.x * kBlock) {
        float r = accum[3 * (size_t)pixel], g = accum[3 * (size_t)pixel + 1], b = accum[3 * (size_t)pixel + 2];
        for (uint32_t s = 0; s < P.batch_samples; ++s) {
            float4 l = radiance[(size_t)s * P.npix + pixel];
            if (P.mode == 1u) {
                l.x = l.x > 0.0f ? sqrtf(l.x) : 0.0f;
                l.y = l.y > 0.0f ? sqrtf(l.y) : 0.0f;
                l.z = l.z > 0.0f ? sqrtf(l.z) : 0.0f;
            }
            r += l.x;
            g += l.y;
            b += l.z;
        }
        accum[3 * (size_t)pixel] = r;
        accum[3 * (size_t)pixel + 1] = g;
        accum[3 * (size_t)pixel + 2] = b;
    }
}

// ------------------------------------------------------------------------------------------------------
// helpers of rt_trace_closest: AoS rays -> wavefront layout, wavefront hits -> rt_hit records
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_pack_rays(const float *__restrict__ rays, uint32_t n, float4 *ray_o,
                                                      float4 *ray_d) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float *r = rays + 6 * (size_t)i;
        ray_o[i] = make_float4(r[0], r[1], r[2], 0.0f);
        ray_d[i] = make_float4(r[3], r[4], r[5], 0.0f);
    }
}

__global__ __launch_bounds__(kBlock) void k_finalize_hits(SceneDev sc, const float4 *__restrict__ ray_o,
                                                          const float4 *__restrict__ ray_d,
                                                          const float4 *__restrict__ hits, uint32_t n,
                                                          uint32_t *__restrict__ out) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        float4 h = hits[i];
        uint32_t leaf = f2u(h.y);
        uint32_t *o = out + 8 * (size_t)i;
        if (leaf == kMissSlot) {
            o[0] = 0u; o[1] = 0xFFFFFFFFu; o[2] = 0u; o[3] = 0u; o[4] = 0u; o[5] = 0u; o[6] = 0u; o[7] = 0u;
            continue;
        }
        Surface s = make_surface<false>(sc, reinterpret_cast<const float4 *>(sc.prims), xyz(ray_o[i]), xyz(ray_d[i]), h.x, leaf,
                                        h.z, h.w);
        o[0] = f2u(h.x);
        o[1] = s.prim_id;
        o[2] = f2u(h.z);
        o[3] = f2u(h.w);
        o[4] = f2u(s.normal.x);
        o[5] = f2u(s.normal.y);
        o[6] = f2u(s.normal.z);
        o[7] = s.front_facing ? 1u : 0u;
    }
}

// ------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------
static inline uint32_t grid_for(uint64_t items, uint32_t max_blocks) {
    uint64_t need = (items + kBlock - 1) / kBlock;
    if (need < 1) need = 1;
    return (uint32_t)(need < max_blocks ? need : max_blocks);
}

void launch_generate(hipStream_t s, uint32_t max_blocks, const RenderParams &P, const PathState &st, uint32_t n_slots) {
    hipLaunchKernelGGL(k_generate, dim3(grid_for(n_slots, max_blocks)), dim3(kBlock), 0, s, P, st, n_slots);
}

void launch_extend_lanes(hipStream_t s, uint32_t grid, const SceneDev &sc, const float4 *ray_o, const float4 *ray_d, float4 *hits,
                         uint32_t n, float tmin, float tmax, const StackSpill &spill, bool chain) {
    constexpr int K = kStackEntriesInLds;
    if (chain) { // one 1024-thread block per CU shares the copy of the top records (see launch_paths_lanes)
        const uint32_t lds_bytes = (uint32_t)(K * 1024 * sizeof(uint2)) + sc.n_chain_top * (uint32_t)sizeof(ChainNode);
        auto kernel = k_extend_lanes<K, true, 1024>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(1024), lds_bytes, s, sc, ray_o, ray_d, hits, n, tmin, tmax, spill);
    } else
        hipLaunchKernelGGL((k_extend_lanes<K, false>), dim3(grid), dim3(kBlock), sc.top_bytes, s, sc, ray_o, ray_d, hits, n, tmin, tmax, spill);
}

bool launch_extend_plain(hipStream_t s, uint32_t grid, const SceneDev &sc, const float4 *ray_o, const float4 *ray_d, float4 *hits,
                         uint32_t n, float tmin, float tmax, ExtendCounters *counters, bool octants) {
    const uint32_t lds_bytes = sc.n_nodes * (octants ? kOctantNodeBytes : kSortedNodeBytes) + sc.n_prims * (uint32_t)sizeof(Prim);
    if (lds_bytes > 150u * 1024u) return false;
    auto kernel = octants ? k_extend_plain<false> : k_extend_plain<true>;
    if (lds_bytes > 48u * 1024u)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), lds_bytes, s, sc, ray_o, ray_d, hits, n, tmin, tmax, counters);
    return true;
}

void launch_extend(hipStream_t s, uint32_t grid, const SceneDev &sc, const float4 *ray_o, const float4 *ray_d,
                   float4 *hits, const QueueView &q, float tmin, float tmax, bool use_bvh, ExtendCounters *counters,
                   uint32_t lds_bytes) {
    dim3 g(grid), b(kBlock);
    if (lds_bytes > 48u * 1024u) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_extend<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_extend<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    }
    if (!use_bvh)
        hipLaunchKernelGGL(k_extend_world, g, b, 0, s, sc, ray_o, ray_d, hits, q, tmin, tmax);
    else if (counters && lds_bytes)
        hipLaunchKernelGGL((k_extend<true, true>), g, b, lds_bytes, s, sc, ray_o, ray_d, hits, q, tmin, tmax, counters);
    else if (counters)
        hipLaunchKernelGGL((k_extend<true, false>), g, b, 0, s, sc, ray_o, ray_d, hits, q, tmin, tmax, counters);
    else if (lds_bytes)
        hipLaunchKernelGGL((k_extend<false, true>), g, b, lds_bytes, s, sc, ray_o, ray_d, hits, q, tmin, tmax, counters);
    else
        hipLaunchKernelGGL((k_extend<false, false>), g, b, 0, s, sc, ray_o, ray_d, hits, q, tmin, tmax, counters);
}

void launch_shade(hipStream_t s, uint32_t grid, const SceneDev &sc, const RenderParams &P, const PathState &st,
                  const QueueView &q, uint32_t *queue_out, uint32_t *counts_out, uint32_t *early_out, bool need_uv,
                  bool has_mix) {
    dim3 g(grid), b(kBlock);
    if (has_mix)
        hipLaunchKernelGGL((k_shade<true, true>), g, b, 0, s, sc, P, st, q, queue_out, counts_out, early_out);
    else if (need_uv)
        hipLaunchKernelGGL((k_shade<true, false>), g, b, 0, s, sc, P, st, q, queue_out, counts_out, early_out);
    else
        hipLaunchKernelGGL((k_shade<false, false>), g, b, 0, s, sc, P, st, q, queue_out, counts_out, early_out);
}

template <bool FIRST>
static void launch_bounce_t(hipStream_t s, uint32_t grid, const SceneDev &sc, const RenderParams &P, const PathState &st,
                            const QueueView &q, uint32_t *queue_out, uint32_t *counts_out, uint32_t *early_out,
                            bool need_uv, uint32_t lds_bytes) {
    dim3 g(grid), b(kBlock);
    if (lds_bytes) {
        lds_bytes = sc.n_nodes * kSortedNodeBytes + sc.n_prims * (uint32_t)sizeof(Prim); // the sorted copy (stage_scene_sorted)
        auto kernel = need_uv ? k_bounce<FIRST, true, true> : k_bounce<FIRST, true, false>;
        if (lds_bytes > 48u * 1024u)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(kernel, g, b, lds_bytes, s, sc, P, st, q, queue_out, counts_out, early_out, StackSpill{});
    } else {
        if (need_uv)
            hipLaunchKernelGGL((k_bounce<FIRST, false, true>), g, b, 0, s, sc, P, st, q, queue_out, counts_out, early_out, StackSpill{});
        else
            hipLaunchKernelGGL((k_bounce<FIRST, false, false>), g, b, 0, s, sc, P, st, q, queue_out, counts_out, early_out, StackSpill{});
    }
}

void launch_bounce(hipStream_t s, uint32_t grid, const SceneDev &sc, const RenderParams &P, const PathState &st,
                   const QueueView &q, uint32_t *queue_out, uint32_t *counts_out, uint32_t *early_out, bool first,
                   bool need_uv, uint32_t lds_bytes) {
    if (first)
        launch_bounce_t<true>(s, grid, sc, P, st, q, queue_out, counts_out, early_out, need_uv, lds_bytes);
    else
        launch_bounce_t<false>(s, grid, sc, P, st, q, queue_out, counts_out, early_out, need_uv, lds_bytes);
}

void launch_paths(hipStream_t s, uint32_t grid, const SceneDev &sc, const RenderParams &P, const PathState &st, uint32_t n_slots,
                  uint32_t cap, uint32_t *queue0, uint32_t *queue1, uint32_t *counts, uint32_t *early, bool need_uv,
                  bool has_mix, uint32_t lds_bytes, bool big_blocks, bool eight_waves) {
    using Kernel = void (*)(SceneDev, RenderParams, PathState, uint32_t, uint32_t, uint32_t *, uint32_t *, uint32_t *, uint32_t *);
    Kernel kernel;
    uint32_t threads = kBlock;
    if (lds_bytes) lds_bytes = sc.n_nodes * kSortedNodeBytes + sc.n_prims * (uint32_t)sizeof(Prim); // the sorted copy (stage_scene_sorted)
    // Tiny scenes (Cornell: 71 nodes, 20 KB): eight copies per node, as long as six 256-thread blocks still fit a CU.
    const uint32_t octant_bytes = sc.n_nodes * kOctantNodeBytes + sc.n_prims * (uint32_t)sizeof(Prim);
    // Scenes with Mix materials use the instances that also interpolate texture coordinates.
    if (lds_bytes && !big_blocks && !has_mix && octant_bytes <= 24u * 1024u) {
        kernel = need_uv ? k_paths<true, true, kBlock, 6, false, true> : k_paths<true, false, kBlock, 6, false, true>;
        lds_bytes = octant_bytes;
    } else if (!lds_bytes && eight_waves) {
        kernel = has_mix ? k_paths<false, true, kBlock, 8, true> : need_uv ? k_paths<false, true, kBlock, 8> : k_paths<false, false, kBlock, 8>;
    } else if (lds_bytes && big_blocks) {
        kernel = has_mix ? k_paths<true, true, 1024, 6, true> : need_uv ? k_paths<true, true, 1024> : k_paths<true, false, 1024>;
        threads = 1024;
    } else if (lds_bytes) {
        kernel = has_mix ? k_paths<true, true, kBlock, 6, true> : need_uv ? k_paths<true, true> : k_paths<true, false>;
    } else {
        kernel = has_mix ? k_paths<false, true, kBlock, 6, true> : need_uv ? k_paths<false, true> : k_paths<false, false>;
    }
    if (lds_bytes > 48u * 1024u)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds_bytes, s, sc, P, st, n_slots, cap, queue0, queue1, counts, early);
}

namespace {
using LanesKernel = void (*)(SceneDev, RenderParams, PathState, uint32_t, uint32_t, const uint32_t *, const uint32_t *, uint32_t *, StackSpill,
                             uint32_t);

// The instance of k_paths_lanes over the walk table for a block size and a register budget.
template <bool FROM_CAMERA, int THREADS, int WAVES>
LanesKernel walk_kernel(bool need_uv, bool has_mix) {
    constexpr int K = kStackEntriesInLds;
    return has_mix   ? k_paths_lanes<true, true, 4, K, FROM_CAMERA, false, THREADS>
           : need_uv ? k_paths_lanes<true, false, WAVES, K, FROM_CAMERA, false, THREADS>
                     : k_paths_lanes<false, false, WAVES, K, FROM_CAMERA, false, THREADS>;
}
template <bool FROM_CAMERA>
LanesKernel walk_kernel(uint32_t threads, int waves, bool need_uv, bool has_mix) {
    if (threads == 1024u) return waves >= 8 ? walk_kernel<FROM_CAMERA, 1024, 8>(need_uv, has_mix) : walk_kernel<FROM_CAMERA, 1024, 4>(need_uv, has_mix);
    if (threads == 768u) return walk_kernel<FROM_CAMERA, 768, 6>(need_uv, has_mix);
    if (threads == 512u)
        return waves >= 8   ? walk_kernel<FROM_CAMERA, 512, 8>(need_uv, has_mix)
               : waves >= 6 ? walk_kernel<FROM_CAMERA, 512, 6>(need_uv, has_mix)
                            : walk_kernel<FROM_CAMERA, 512, 4>(need_uv, has_mix);
    return waves >= 6 ? walk_kernel<FROM_CAMERA, kBlock, 6>(need_uv, has_mix) : walk_kernel<FROM_CAMERA, kBlock, 4>(need_uv, has_mix);
}
} // namespace

void launch_paths_lanes(hipStream_t s, uint32_t grid, const SceneDev &sc, const RenderParams &P, const PathState &st,
                        uint32_t n_slots, uint32_t cap, uint32_t *queue, uint32_t *counts, uint32_t *early, bool need_uv, bool has_mix,
                        int waves, const StackSpill &spill, bool camera_rays_in_lockstep, bool chain, uint32_t walk_threads,
                        uint32_t walk_grid) {
    constexpr int K = kStackEntriesInLds;
    if (!chain) { // the stackless walk: no stack memory; LDS holds the treetop of the walk table
        if (camera_rays_in_lockstep) {
            const QueueView all{nullptr, nullptr, cap, n_slots};
#ifdef RT_FIRST_BOUNCE_PREORDER
            auto first = has_mix ? k_bounce<true, false, true, true, 0> : need_uv ? k_bounce<true, false, true, false, 0> : k_bounce<true, false, false, false, 0>;
            hipLaunchKernelGGL(first, dim3(grid), dim3(kBlock), 0, s, sc, P, st, all, queue, counts + grid, early + grid, spill);
#else
            // six 256-thread blocks to a CU: 24 KB of treetop each
            SceneDev first_sc = sc;
            first_sc.top_bytes = sc.top_bytes < 24u * 1024u ? sc.top_bytes : 24u * 1024u;
            auto first = has_mix ? k_bounce<true, false, true, true, -1> : need_uv ? k_bounce<true, false, true, false, -1> : k_bounce<true, false, false, false, -1>;
            hipLaunchKernelGGL(first, dim3(grid), dim3(kBlock), first_sc.top_bytes, s, first_sc, P, st, all, queue, counts + grid, early + grid, spill);
#endif
            if (P.max_depth < 2u) return;
        }
        LanesKernel kernel = camera_rays_in_lockstep ? walk_kernel<false>(walk_threads, waves, need_uv, has_mix)
                                                     : walk_kernel<true>(walk_threads, waves, need_uv, has_mix);
        if (sc.top_bytes > 48u * 1024u)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sc.top_bytes);
        uint32_t *ray_counts = counts + (size_t)(camera_rays_in_lockstep ? 2u : 1u) * grid;
        const uint32_t lds_bytes = sc.top_bytes + (sc.prims_in_lds ? sc.n_prims * (uint32_t)sizeof(Prim) : 0u);
        if (lds_bytes > 48u * 1024u)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(kernel, dim3(walk_grid), dim3(walk_threads), lds_bytes, s, sc, P, st, n_slots, cap, queue, counts + grid,
                           ray_counts, spill, grid / walk_grid);
        return;
    }
    if (camera_rays_in_lockstep) { // bounce 0 by the per-bounce kernel: survivors into `queue`, rows 1 of the tables
        const QueueView all{nullptr, nullptr, cap, n_slots};
        auto first = has_mix ? k_bounce<true, false, true, true, K> : need_uv ? k_bounce<true, false, true, false, K> : k_bounce<true, false, false, false, K>;
        hipLaunchKernelGGL(first, dim3(grid), dim3(kBlock), (uint32_t)(K * kBlock * sizeof(uint2)), s, sc, P, st, all, queue, counts + grid,
                           early + grid, spill);
        if (P.max_depth < 2u) return;
    }
    LanesKernel kernel;
#define RT_LANES_KERNEL(FROM_CAMERA)                                                                                             \
    (walk_threads == 1024u ? (has_mix ? k_paths_lanes<true, true, 4, K, FROM_CAMERA, true, 1024>                                   \
                              : need_uv ? k_paths_lanes<true, false, 4, K, FROM_CAMERA, true, 1024>                                \
                                        : k_paths_lanes<false, false, 4, K, FROM_CAMERA, true, 1024>)                              \
     : walk_threads == 512u ? (has_mix ? k_paths_lanes<true, true, 4, K, FROM_CAMERA, true, 512>                                   \
                               : need_uv ? k_paths_lanes<true, false, 4, K, FROM_CAMERA, true, 512>                                \
                                         : k_paths_lanes<false, false, 4, K, FROM_CAMERA, true, 512>)                              \
     : has_mix    ? k_paths_lanes<true, true, 4, K, FROM_CAMERA>                                                                   \
     : waves >= 8 ? (need_uv ? k_paths_lanes<true, false, 8, K, FROM_CAMERA> : k_paths_lanes<false, false, 8, K, FROM_CAMERA>)     \
     : waves >= 6 ? (need_uv ? k_paths_lanes<true, false, 6, K, FROM_CAMERA> : k_paths_lanes<false, false, 6, K, FROM_CAMERA>)     \
     : waves >= 5 ? (need_uv ? k_paths_lanes<true, false, 5, K, FROM_CAMERA> : k_paths_lanes<false, false, 5, K, FROM_CAMERA>)     \
                  : (need_uv ? k_paths_lanes<true, false, 4, K, FROM_CAMERA> : k_paths_lanes<false, false, 4, K, FROM_CAMERA>))
    if (camera_rays_in_lockstep)
        kernel = RT_LANES_KERNEL(false);
    else
        kernel = RT_LANES_KERNEL(true);
#undef RT_LANES_KERNEL
    const uint32_t threads = (walk_threads == 1024u || walk_threads == 512u) ? walk_threads : kBlock;
    const uint32_t lds_bytes = (uint32_t)(K * threads * sizeof(uint2)) + sc.n_chain_top * (uint32_t)sizeof(ChainNode);
    if (lds_bytes > 48u * 1024u)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    // rays beyond those the tables already hold: bounce 0's survivors are row 1, so the lanes' own count is row 2
    uint32_t *ray_counts = counts + (size_t)(camera_rays_in_lockstep ? 2u : 1u) * grid;
    hipLaunchKernelGGL(kernel, dim3(walk_grid), dim3(threads), lds_bytes, s, sc, P, st, n_slots, cap, queue, counts + grid, ray_counts, spill,
                       grid / walk_grid);
}

// ------------------------------------------------------------------------------------------------------
// Scratch of the allocation-free closest-hit entry point.  The array's name is only a host-side handle: host code
// must go through the address hipGetSymbolAddress returns for the device that is current, and each device has
// its own.  The contents are whatever the last call on that device left; every call writes the slots it reads.
// ------------------------------------------------------------------------------------------------------
static __device__ __align__(256) float4 scratch[3 * kNoAllocMaxRays];

float4 *noalloc_scratch_of_current_device(const char **error) {
    constexpr int kMaxDevices = 64;
    static float4 *ptr_of_device[kMaxDevices] = {};
    static std::mutex mutex;
    int device = -1;
    hipError_t e = hipGetDevice(&device);
    if (e == hipSuccess && (device < 0 || device >= kMaxDevices)) e = hipErrorInvalidDevice;
    if (e != hipSuccess) {
        if (error) *error = hipGetErrorString(e);
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(mutex);
    if (!ptr_of_device[device]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            if (error) *error = hipGetErrorString(e);
            return nullptr;
        }
        ptr_of_device[device] = static_cast<float4 *>(p);
    }
    return ptr_of_device[device];
}

void launch_sum_counts(hipStream_t s, const uint32_t *counts, const uint32_t *early, uint32_t n_blocks, uint32_t n_rows,
                       unsigned long long *totals) {
    hipLaunchKernelGGL(k_sum_counts, dim3(n_rows), dim3(kBlock), 0, s, counts, early, n_blocks, totals);
}

void launch_accumulate(hipStream_t s, uint32_t max_blocks, const RenderParams &P, const float4 *radiance, float *accum) {
    hipLaunchKernelGGL(k_accumulate, dim3(grid_for(P.npix, max_blocks)), dim3(kBlock), 0, s, P, radiance, accum);
}

void launch_pack_rays(hipStream_t s, uint32_t max_blocks, const float *rays, uint32_t n, float4 *ray_o, float4 *ray_d) {
    hipLaunchKernelGGL(k_pack_rays, dim3(grid_for(n, max_blocks)), dim3(kBlock), 0, s, rays, n, ray_o, ray_d);
}

void launch_finalize_hits(hipStream_t s, uint32_t max_blocks, const SceneDev &sc, const float4 *ray_o,
                          const float4 *ray_d, const float4 *hits, uint32_t n, void *out) {
    hipLaunchKernelGGL(k_finalize_hits, dim3(grid_for(n, max_blocks)), dim3(kBlock), 0, s, sc, ray_o, ray_d, hits, n,
                       reinterpret_cast<uint32_t *>(out));
}

} // namespace rt
