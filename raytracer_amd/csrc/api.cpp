// api.cpp — implementation of the C ABI declared in include/rt_mi355x.h.
//
// Host-side driver of the wavefront pipeline: it owns the HBM workspace, enqueues generate / extend / shade /
// accumulate on the context's stream and never reads a queue length back while a batch is in flight.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "bvh_build.h"
#include "kernels.h"
#include "rt_device.h"
#include "rt_math.h"
#include "sample_split.h"
#include "scene_build.h"

#define RT_API extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_error;

int set_error(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

} // namespace

namespace rt {
void set_last_error(const std::string &message) { g_error = message; }
} // namespace rt

namespace {

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return set_error(e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_DEVICE, "%s: %s", #expr, \
                             hipGetErrorString(e_));                                                       \
    } while (0)

template <class T> struct DevBuf {
    T *ptr = nullptr;
    size_t cap = 0; // elements
    int ensure(size_t n) {
        if (n <= cap) return RT_OK;
        if (ptr) {
            HIP_TRY(hipFree(ptr));
            ptr = nullptr;
            cap = 0;
        }
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ptr), n * sizeof(T)));
        cap = n;
        return RT_OK;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

enum KernelKind { K_GENERATE = 0, K_EXTEND = 1, K_SHADE = 2, K_ACCUMULATE = 3, K_BOUNCE = 4, K_BOUNCE0 = 5, K_KINDS = 6 };

} // namespace

struct rt_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    uint32_t flags = 0;
    uint32_t max_blocks = 2048;       // grid of the helper kernels: 8 blocks per CU
    uint32_t cus = 256;
    uint32_t blocks_per_cu = 0;       // render grid, blocks per CU; 0 = chosen per scene (see rt_render_device)
    uint64_t max_paths = 1ull << 26;
    uint32_t timed_bounces = 0xFFFFFFFFu;
    uint32_t pipeline = RT_PIPELINE_AUTO;
    // LANES knobs, 0 = chosen per scene (environment variables RT_LANES_TRAVERSAL, RT_LANES_FIRST, RT_LANES_WAVES)
    int lanes_traversal = 0; // +1 chain records, -1 stackless walk
    int lanes_first = 0;     // +1 bounce 0 by the per-bounce kernel in lockstep, -1 lanes from the camera ray on
    int lanes_waves = 0;     // waves per SIMD the kernel's registers are budgeted for
    // launch shape of the lanes kernel over the walk table, 0 = default (RT_WALK_THREADS, RT_WALK_BLOCKS_PER_CU, RT_WALK_TOP_KB)
    uint32_t walk_threads = 0, walk_blocks_per_cu = 0;
    int64_t walk_top_bytes = -1;
    // the same for the chain walk (RT_CHAIN_THREADS, RT_CHAIN_TOP_RECORDS)
    uint32_t chain_threads = 0;
    int64_t chain_top_records = -1;
    int trace_traversal = 0; // RT_TRACE_TRAVERSAL: closest-hit queries by the render kernels' LDS walks (1 sorted, 2 octants)
    float4 *noalloc_scratch = nullptr; // this device's copy of the allocation-free entry point's scratch array

    // wavefront workspace
    DevBuf<float4> ray_o, ray_d, hit, beta, radiance;
    DevBuf<uint32_t> queue[2];
    DevBuf<uint32_t> counts;              // [max_depth + 1][grid] survivors per block and bounce
    DevBuf<uint32_t> early;               // [max_depth + 1][grid] early misses per block and bounce
    DevBuf<unsigned long long> totals;    // [max_depth + 1] per-bounce sums of `counts`
    DevBuf<rt::ExtendCounters> counters;
    DevBuf<uint2> stack_spill;            // [tree depth][grid threads] traversal-stack entries beyond the LDS part
    DevBuf<float> accum;     // rt_render's device image
    DevBuf<float> ray_stage; // rt_trace_closest's device copies
    DevBuf<uint32_t> hit_stage;
    unsigned long long *host_counts = nullptr; // pinned
    size_t host_counts_cap = 0;

    // timing
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    std::vector<hipEvent_t> ev_pool;
    struct Span {
        int kind;
        size_t a, b;
    };
    std::vector<Span> spans;
    size_t ev_used = 0;
};

struct rt_scene {
    int device = 0;
    rt::SceneDev dev{};
    std::vector<void *> blocks; // device allocations the scene owns
    // The chain records and the walk table are derived from the node table when a traversal first asks for them
    // (ensure_chain / ensure_walk): dev.chain / dev.walk_nodes stay nullptr until then.
    std::mutex derived_mutex;
    uint32_t walk_region_bytes = 0; // bytes at the start of the walk table any prefix of which is a treetop
    uint32_t chain_top_made_for = 0; // record count dev.chain_top was made for (ensure_chain_top)
    std::vector<uint32_t> prim_of_slot;
    rt_scene_info info{};
    bool need_uv = false;
    bool has_mix = false;
};

namespace {

int ensure_paths(rt_context *c, size_t n) {
    if (int rc = c->ray_o.ensure(n)) return rc;
    if (int rc = c->ray_d.ensure(n)) return rc;
    if (int rc = c->hit.ensure(n)) return rc;
    return RT_OK;
}

int ensure_render(rt_context *c, size_t n, size_t queue_len, uint32_t max_depth, uint32_t grid) {
    if (int rc = ensure_paths(c, n)) return rc;
    if (int rc = c->beta.ensure(n)) return rc;
    if (int rc = c->radiance.ensure(n)) return rc;
    if (int rc = c->queue[0].ensure(queue_len)) return rc;
    if (int rc = c->queue[1].ensure(queue_len)) return rc;
    if (int rc = c->counts.ensure(((size_t)max_depth + 1) * grid)) return rc;
    if (int rc = c->early.ensure(((size_t)max_depth + 1) * grid)) return rc;
    if (int rc = c->totals.ensure((size_t)max_depth + 1)) return rc;
    if (int rc = c->counters.ensure(1)) return rc;
    return RT_OK;
}

int ensure_host_counts(rt_context *c, size_t n) {
    if (n <= c->host_counts_cap) return RT_OK;
    if (c->host_counts) HIP_TRY(hipHostFree(c->host_counts));
    c->host_counts = nullptr;
    c->host_counts_cap = 0;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->host_counts), n * sizeof(unsigned long long),
                          hipHostMallocDefault));
    c->host_counts_cap = n;
    return RT_OK;
}

int timed_begin(rt_context *c, int kind, bool on = true) {
    if (!(c->flags & RT_CTX_TIMING) || !on) return RT_OK;
    while (c->ev_pool.size() < c->ev_used + 2) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        c->ev_pool.push_back(e);
    }
    HIP_TRY(hipEventRecord(c->ev_pool[c->ev_used], c->stream));
    c->spans.push_back({kind, c->ev_used, c->ev_used + 1});
    return RT_OK;
}

int timed_end(rt_context *c, bool on = true) {
    if (!(c->flags & RT_CTX_TIMING) || !on) return RT_OK;
    HIP_TRY(hipEventRecord(c->ev_pool[c->ev_used + 1], c->stream));
    c->ev_used += 2;
    return RT_OK;
}

int collect_timing(rt_context *c, rt_stats *st) {
    if (!(c->flags & RT_CTX_TIMING)) return RT_OK;
    double sums[K_KINDS] = {0, 0, 0, 0, 0, 0};
    for (const auto &s : c->spans) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_pool[s.a], c->ev_pool[s.b]));
        sums[s.kind] += ms;
    }
    st->generate_ms = sums[K_GENERATE];
    st->extend_ms = sums[K_EXTEND];
    st->shade_ms = sums[K_SHADE];
    st->accumulate_ms = sums[K_ACCUMULATE];
    st->bounce_ms = sums[K_BOUNCE] + sums[K_BOUNCE0];
    st->first_bounce_ms = sums[K_BOUNCE0];
    return RT_OK;
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

template <class T> int upload(rt_scene *s, const std::vector<T> &v, const T **dst, uint64_t &bytes) {
    *dst = nullptr;
    size_t n = v.size() ? v.size() : 1;
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, n * sizeof(T)));
    s->blocks.push_back(p);
    if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<const T *>(p);
    bytes += n * sizeof(T);
    return RT_OK;
}

} // namespace

namespace {

// LDS bytes every block of a launch stages the walk table's treetop in: blocks of `threads` threads, `blocks_per_cu` of
// them resident on a CU with 160 KB of LDS (2 KB per block left for the kernels' own shared variables and the
// allocation granularity).
uint32_t top_bytes_for(const rt_scene *s, const rt_context *c, uint32_t blocks_per_cu) {
    const uint32_t fit = ((160u * 1024u) / std::max(1u, blocks_per_cu) - 2048u) & ~31u;
    uint32_t want = c->walk_top_bytes >= 0 ? (uint32_t)c->walk_top_bytes : fit;
    want = std::min(want, fit) & ~31u;
    return std::min(want, s->walk_region_bytes);
}

// The chain records (ChainNode) of a scene, built on the device from its node table the first time a traversal asks.
int ensure_chain(rt_context *c, const rt_scene *cs) {
    rt_scene *s = const_cast<rt_scene *>(cs);
    std::lock_guard<std::mutex> lock(s->derived_mutex);
    if (s->dev.chain) return RT_OK;
    if (s->dev.n_prims > rt::kRefIndex) return set_error(RT_ERR_UNSUPPORTED, "the chain records hold 30-bit primitive slots");
    void *p = nullptr;
    const size_t bytes = (size_t)std::max<uint32_t>(1u, s->dev.n_prims - 1u) * sizeof(rt::ChainNode);
    HIP_TRY(hipMalloc(&p, bytes));
    rt::build_chain_nodes(c->stream, s->dev.nodes, s->dev.n_nodes, static_cast<rt::ChainNode *>(p));
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
        (void)hipFree(p);
        return set_error(RT_ERR_DEVICE, "building the BVH node records failed");
    }
    s->blocks.push_back(p);
    s->dev.chain = static_cast<const rt::ChainNode *>(p);
    s->info.device_bytes += bytes;
    return RT_OK;
}

// Copies of the `want` chain records nearest the root for the lanes kernel to keep in LDS: breadth first over the records a
// walk visits (a record that holds a pair of grandchildren is followed by the grandchildren's records, else by the
// child's), child references that point into the copy tagged with kRefTop.  A copy serves exactly the count it was made
// for (its tagged references name slots up to that count), so it is remade when a launch asks for another count.
int ensure_chain_top(rt_context *c, const rt_scene *cs, uint32_t want) {
    rt_scene *s = const_cast<rt_scene *>(cs);
    std::lock_guard<std::mutex> lock(s->derived_mutex);
    want = std::min(want, s->dev.n_prims > 1u ? s->dev.n_prims - 1u : 0u);
    if (s->chain_top_made_for == want) return RT_OK;
    // An older copy stays allocated until the scene is destroyed (a launch on another stream may still read it); a
    // launch shape, and with it the count, changes only when a tuning knob does.
    s->dev.chain_top = nullptr;
    s->dev.n_chain_top = 0;
    s->chain_top_made_for = want;
    if (want == 0) return RT_OK;
    std::vector<uint32_t> order{0u};
    std::vector<rt::ChainNode> recs;
    for (size_t i = 0; i < order.size(); ++i) {
        rt::ChainNode r;
        HIP_TRY(hipMemcpy(&r, s->dev.chain + order[i], sizeof r, hipMemcpyDeviceToHost));
        recs.push_back(r);
        const uint32_t next[4] = {(r.meta & (1u << 16)) ? r.ll : r.left, (r.meta & (1u << 16)) ? r.lr : rt::kRefNone,
                                  (r.meta & (1u << 17)) ? r.rl : r.right, (r.meta & (1u << 17)) ? r.rr : rt::kRefNone};
        for (uint32_t ref : next)
            if (!(ref & rt::kRefLeaf) && order.size() < want) order.push_back(ref);
    }
    std::vector<std::pair<uint32_t, uint32_t>> slot_of(order.size());
    for (size_t k = 0; k < order.size(); ++k) slot_of[k] = {order[k], (uint32_t)k};
    std::sort(slot_of.begin(), slot_of.end());
    auto patch = [&](uint32_t &ref) {
        if (ref & rt::kRefLeaf) return;
        auto it = std::lower_bound(slot_of.begin(), slot_of.end(), std::make_pair(ref, 0u));
        if (it != slot_of.end() && it->first == ref) ref = rt::kRefTop | it->second;
    };
    for (rt::ChainNode &r : recs) {
        patch(r.left), patch(r.right);
        if (r.meta & (1u << 16)) patch(r.ll), patch(r.lr);
        if (r.meta & (1u << 17)) patch(r.rl), patch(r.rr);
    }
    void *t = nullptr;
    HIP_TRY(hipMalloc(&t, recs.size() * sizeof(rt::ChainNode)));
    if (hipMemcpy(t, recs.data(), recs.size() * sizeof(rt::ChainNode), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(t);
        return set_error(RT_ERR_DEVICE, "uploading the top chain records failed");
    }
    s->blocks.push_back(t);
    s->dev.chain_top = static_cast<const rt::ChainNode *>(t);
    s->dev.n_chain_top = (uint32_t)recs.size();
    return RT_OK;
}

// The walk table (WalkNode) of a scene, built on the host from a copy of its node table the first time a traversal asks.
int ensure_walk(rt_context *c, const rt_scene *cs) {
    rt_scene *s = const_cast<rt_scene *>(cs);
    std::lock_guard<std::mutex> lock(s->derived_mutex);
    if (s->dev.walk_nodes) return RT_OK;
    try {
        std::vector<rt::Node> nodes(s->dev.n_nodes);
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(nodes.data(), s->dev.nodes, nodes.size() * sizeof(rt::Node), hipMemcpyDeviceToHost));
        std::vector<rt::WalkNode> table;
        size_t region = 0;
        std::string err;
        if (!rt::build_walk_table(nodes.data(), nodes.size(), (158u * 1024u) / sizeof(rt::WalkNode), table, &region, err))
            return set_error(RT_ERR_UNSUPPORTED, "%s", err.c_str());
        void *p = nullptr;
        HIP_TRY(hipMalloc(&p, table.size() * sizeof(rt::WalkNode)));
        if (hipMemcpy(p, table.data(), table.size() * sizeof(rt::WalkNode), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(p);
            return set_error(RT_ERR_DEVICE, "uploading the walk table failed");
        }
        s->blocks.push_back(p);
        s->walk_region_bytes = (uint32_t)(region * sizeof(rt::WalkNode));
        s->dev.walk_nodes = static_cast<const rt::WalkNode *>(p);
        s->info.device_bytes += table.size() * sizeof(rt::WalkNode);
    } catch (const std::bad_alloc &) {
        return set_error(RT_ERR_OUT_OF_MEMORY, "out of host memory while building the walk table");
    }
    return RT_OK;
}

} // namespace

// ------------------------------------------------------------------------------------------------------
RT_API const char *rt_last_error(void) { return g_error.c_str(); }

RT_API int rt_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return set_error(RT_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

RT_API int rt_context_create(int device, void *stream, uint32_t flags, rt_context **out) {
    if (!out) return set_error(RT_ERR_INVALID_ARGUMENT, "rt_context_create: null out pointer");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return set_error(RT_ERR_NO_DEVICE, "no HIP device is visible; this library has no CPU path");
    if (device < 0 || device >= n) return set_error(RT_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, n - 1);
    DeviceGuard guard(device);
    if (!guard.ok) return set_error(RT_ERR_DEVICE, "hipSetDevice(%d) failed", device);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    rt_context *c = new (std::nothrow) rt_context();
    if (!c) return set_error(RT_ERR_OUT_OF_MEMORY, "out of host memory");
    c->device = device;
    c->flags = flags;
    if (const char *e = std::getenv("RT_BLOCKS_PER_CU")) { // tuning knob for experiments
        long v = std::strtol(e, nullptr, 10);
        if (v >= 1 && v <= 64) c->blocks_per_cu = (uint32_t)v;
    }
    if (const char *e = std::getenv("RT_LANES_TRAVERSAL"))
        c->lanes_traversal = !std::strcmp(e, "chain") ? 1 : !std::strcmp(e, "walk") ? -1 : 0;
    if (const char *e = std::getenv("RT_LANES_FIRST")) c->lanes_first = !std::strcmp(e, "lockstep") ? 1 : !std::strcmp(e, "lanes") ? -1 : 0;
    if (const char *e = std::getenv("RT_LANES_WAVES")) {
        long v = std::strtol(e, nullptr, 10);
        if (v >= 4 && v <= 8 && v != 7) c->lanes_waves = (int)v;
    }
    if (const char *e = std::getenv("RT_WALK_THREADS")) {
        long v = std::strtol(e, nullptr, 10);
        if (v == 256 || v == 512 || v == 768 || v == 1024) c->walk_threads = (uint32_t)v;
    }
    if (const char *e = std::getenv("RT_WALK_BLOCKS_PER_CU")) {
        long v = std::strtol(e, nullptr, 10);
        if (v >= 1 && v <= 8) c->walk_blocks_per_cu = (uint32_t)v;
    }
    if (const char *e = std::getenv("RT_TRACE_TRAVERSAL")) c->trace_traversal = !std::strcmp(e, "sorted") ? 1 : !std::strcmp(e, "octants") ? 2 : 0;
    if (const char *e = std::getenv("RT_CHAIN_THREADS")) {
        long v = std::strtol(e, nullptr, 10);
        if (v == 256 || v == 512 || v == 1024) c->chain_threads = (uint32_t)v;
    }
    if (const char *e = std::getenv("RT_CHAIN_TOP_RECORDS")) {
        long v = std::strtol(e, nullptr, 10);
        if (v >= 0 && v <= 1200) c->chain_top_records = v;
    }
    if (const char *e = std::getenv("RT_WALK_TOP_KB")) {
        long v = std::strtol(e, nullptr, 10);
        if (v >= 0 && v <= 150) c->walk_top_bytes = v * 1024;
    }
    c->cus = (uint32_t)std::max(1, prop.multiProcessorCount);
    c->max_blocks = c->cus * 8u;
    if (stream) {
        c->stream = static_cast<hipStream_t>(stream);
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            return set_error(RT_ERR_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e));
        }
        c->owns_stream = true;
    }
    if (hipEventCreate(&c->ev_begin) != hipSuccess || hipEventCreate(&c->ev_end) != hipSuccess) {
        rt_context_destroy(c);
        return set_error(RT_ERR_DEVICE, "hipEventCreate failed");
    }
    // Resolved here, with the context's device current, so that rt_trace_closest_noalloc itself has no set-up to do.
    const char *why = nullptr;
    c->noalloc_scratch = rt::noalloc_scratch_of_current_device(&why);
    if (!c->noalloc_scratch) {
        rt_context_destroy(c);
        return set_error(RT_ERR_DEVICE, "hipGetSymbolAddress(scratch) on device %d: %s", device, why ? why : "no address");
    }
    *out = c;
    return RT_OK;
}

RT_API int rt_context_destroy(rt_context *c) {
    if (!c) return RT_OK;
    DeviceGuard guard(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->ray_o.release();
    c->ray_d.release();
    c->hit.release();
    c->beta.release();
    c->radiance.release();
    c->queue[0].release();
    c->queue[1].release();
    c->counts.release();
    c->early.release();
    c->totals.release();
    c->counters.release();
    c->stack_spill.release();
    c->accum.release();
    c->ray_stage.release();
    c->hit_stage.release();
    if (c->host_counts) (void)hipHostFree(c->host_counts);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_end) (void)hipEventDestroy(c->ev_end);
    if (c->owns_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RT_OK;
}

RT_API int rt_context_synchronize(rt_context *c) {
    if (!c) return set_error(RT_ERR_INVALID_ARGUMENT, "null context");
    DeviceGuard guard(c->device);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

RT_API int rt_context_set_pipeline(rt_context *c, uint32_t pipeline) {
    if (!c) return set_error(RT_ERR_INVALID_ARGUMENT, "null context");
    if (pipeline > RT_PIPELINE_LANES) return set_error(RT_ERR_INVALID_ARGUMENT, "unknown pipeline %u", pipeline);
    c->pipeline = pipeline;
    return RT_OK;
}

RT_API int rt_context_set_timed_bounces(rt_context *c, uint32_t bounces) {
    if (!c) return set_error(RT_ERR_INVALID_ARGUMENT, "null context");
    c->timed_bounces = bounces ? bounces : 0xFFFFFFFFu;
    return RT_OK;
}

RT_API int rt_context_set_max_paths(rt_context *c, uint64_t max_paths) {
    if (!c) return set_error(RT_ERR_INVALID_ARGUMENT, "null context");
    c->max_paths = max_paths ? max_paths : (1ull << 26);
    return RT_OK;
}

// ------------------------------------------------------------------------------------------------------
RT_API int rt_scene_create(rt_context *c, const rt_scene_desc *desc, rt_scene **out) {
    if (!c || !desc || !out) return set_error(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: null argument");
    *out = nullptr;
    DeviceGuard guard(c->device);
    rt_scene *s = new (std::nothrow) rt_scene();
    if (!s) return set_error(RT_ERR_OUT_OF_MEMORY, "out of host memory");
    s->device = c->device;

    // Scenes of at least RT_DEVICE_BUILD_MIN_PRIMS primitives (default 65536) get their tree built on the GPU
    // (bvh_build.hip); smaller ones by the host builder.  Both produce the same tree.
    uint64_t device_min = 1ull << 16;
    if (const char *e = std::getenv("RT_DEVICE_BUILD_MIN_PRIMS")) device_min = std::strtoull(e, nullptr, 10);
    rt::Node *d_nodes = nullptr;
    uint64_t n_nodes_device = 0;
    rt::ExternalTreeBuilder on_device = [&](const rt::BuildBox *boxes, uint32_t n, uint32_t sphere_begin, uint32_t sphere_end,
                                            std::vector<uint32_t> &order, uint32_t &depth, std::string &e) -> int {
        void *p = nullptr;
        n_nodes_device = 2ull * n - 1;
        if (hipMalloc(&p, n_nodes_device * sizeof(rt::Node)) != hipSuccess) {
            e = "out of device memory for the BVH nodes";
            return RT_ERR_OUT_OF_MEMORY;
        }
        s->blocks.push_back(p);
        d_nodes = static_cast<rt::Node *>(p);
        return rt::build_bvh_device(c->stream, boxes, n, sphere_begin, sphere_end, d_nodes, order.data(), &depth, e);
    };

    rt::HostScene hs;
    std::string err;
    int rc;
    try {
        rc = rt::build_host_scene(desc, hs, err, &on_device, device_min);
    } catch (const std::bad_alloc &) {
        rt_scene_destroy(s);
        return set_error(RT_ERR_OUT_OF_MEMORY, "out of host memory while building the scene");
    } catch (const std::exception &e) {
        rt_scene_destroy(s);
        return set_error(RT_ERR_INVALID_ARGUMENT, "scene build failed: %s", e.what());
    }
    if (rc) {
        rt_scene_destroy(s);
        return set_error(rc, "%s", err.c_str());
    }

    uint64_t bytes = 0;
    rt::Node root{};
    if (d_nodes) {
        s->dev.nodes = d_nodes;
        bytes += n_nodes_device * sizeof(rt::Node);
        if (hipMemcpy(&root, d_nodes, sizeof root, hipMemcpyDeviceToHost) != hipSuccess) rc = set_error(RT_ERR_DEVICE, "reading the root node failed");
    } else {
        root = hs.nodes[0];
        rc = upload(s, hs.nodes, &s->dev.nodes, bytes);
    }
    if (!rc) rc = upload(s, hs.prims, &s->dev.prims, bytes);
    if (!rc) rc = upload(s, hs.tri_shade, &s->dev.tri_shade, bytes);
    if (!rc) rc = upload(s, hs.materials, &s->dev.materials, bytes);
    if (!rc) rc = upload(s, hs.textures, &s->dev.textures, bytes);
    if (!rc) rc = upload(s, hs.texels, &s->dev.texels, bytes);
    if (!rc) rc = upload(s, hs.slot_of_prim, &s->dev.slot_of_prim, bytes);
    if (rc) {
        rt_scene_destroy(s);
        return rc;
    }
    s->dev.n_nodes = d_nodes ? (uint32_t)n_nodes_device : (uint32_t)hs.nodes.size();
    s->dev.n_prims = (uint32_t)hs.prims.size();
    s->dev.depth = hs.depth;
    s->dev.root_ref = root.leaf == rt::kInterior ? 0u
                                                 : (rt::kRefLeaf | ((root.leaf & rt::kSphereBit) ? rt::kRefSphere : 0u) | (root.leaf & ~rt::kSphereBit));
    s->dev.n_materials = (uint32_t)hs.materials.size();
    s->dev.n_textures = (uint32_t)hs.textures.size();
    s->need_uv = hs.need_uv;
    s->has_mix = hs.has_mix;
    s->prim_of_slot.swap(hs.prim_of_slot);
    s->info.num_triangles = hs.n_triangles;
    s->info.num_spheres = hs.n_spheres;
    s->info.num_nodes = s->dev.n_nodes;
    s->info.depth = hs.depth;
    root.get_box(s->info.bounds_min, s->info.bounds_max);
    s->info.device_bytes = bytes;
    *out = s;
    return RT_OK;
}

RT_API int rt_scene_destroy(rt_scene *s) {
    if (!s) return RT_OK;
    DeviceGuard guard(s->device);
    for (void *p : s->blocks)
        if (p) (void)hipFree(p);
    delete s;
    return RT_OK;
}

RT_API int rt_scene_get_info(const rt_scene *s, rt_scene_info *out) {
    if (!s || !out) return set_error(RT_ERR_INVALID_ARGUMENT, "rt_scene_get_info: null argument");
    *out = s->info;
    return RT_OK;
}

RT_API int rt_scene_export_bvh(const rt_scene *s, float *boxes, int32_t *leaf_prim, uint32_t *skip) {
    if (!s || !boxes || !leaf_prim || !skip) return set_error(RT_ERR_INVALID_ARGUMENT, "rt_scene_export_bvh: null argument");
    DeviceGuard guard(s->device);
    std::vector<rt::Node> nodes(s->dev.n_nodes);
    HIP_TRY(hipMemcpy(nodes.data(), s->dev.nodes, nodes.size() * sizeof(rt::Node), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nodes.size(); ++i) {
        nodes[i].get_box(boxes + 6 * i, boxes + 6 * i + 3);
        skip[i] = nodes[i].skip;
        leaf_prim[i] = nodes[i].leaf == rt::kInterior ? -1 : (int32_t)s->prim_of_slot[nodes[i].leaf & ~rt::kSphereBit];
    }
    return RT_OK;
}

// ------------------------------------------------------------------------------------------------------
// Camera (src/camera.rs:20-64), host only
// ------------------------------------------------------------------------------------------------------
RT_API int rt_camera_settings_default(rt_camera_settings *s) {
    if (!s) return set_error(RT_ERR_INVALID_ARGUMENT, "null argument");
    std::memset(s, 0, sizeof *s);
    s->y_fov = 80.0f * (3.14159265358979323846f / 180.0f); // 80.0f32.to_radians()
    s->z_near = 0.0001f;
    s->z_far = std::numeric_limits<float>::infinity();
    s->transform[0] = s->transform[4] = s->transform[8] = 1.0f;
    s->focus_dist = 10.0f;
    s->defocus_angle = 0.0f;
    return RT_OK;
}

RT_API int rt_camera_from_settings(const rt_camera_settings *s, uint32_t width, uint32_t height, rt_camera *out) {
    using namespace rt;
    if (!s || !out) return set_error(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (width == 0 || height == 0) return set_error(RT_ERR_INVALID_ARGUMENT, "image size must be non-zero");
    auto xform = [&](V3 p) { // glam Affine3A::transform_point3a
        const float *m = s->transform;
        V3 r = V3{m[0], m[1], m[2]} * p.x;
        r = r + V3{m[3], m[4], m[5]} * p.y;
        r = r + V3{m[6], m[7], m[8]} * p.z;
        return r + V3{m[9], m[10], m[11]};
    };
    const V3 v_up = V3{0.0f, 1.0f, 0.0f};
    const float theta = s->y_fov;
    const float h = tanf(theta / 2.0f);
    const float viewport_height = 2.0f * h * s->focus_dist;
    const float viewport_width = viewport_height * ((float)width / (float)height);
    const V3 look_from = xform(V3{0.0f, 0.0f, 0.0f});
    const V3 look_at = xform(V3{0.0f, 0.0f, -1.0f});
    const V3 w = normalize(look_from - look_at);
    const V3 u = normalize(cross(v_up, w));
    const V3 v = cross(w, u);
    const V3 viewport_u = u * viewport_width;
    const V3 viewport_v = (-v) * viewport_height;
    const V3 pdu = viewport_u / (float)width;
    const V3 pdv = viewport_v / (float)height;
    const V3 upper_left = ((look_from - w * s->focus_dist) - viewport_u / 2.0f) - viewport_v / 2.0f;
    const V3 p00 = upper_left + (pdu + pdv) * 0.5f;
    const float defocus_radius = s->focus_dist * tanf((s->defocus_angle / 2.0f) * (3.14159265358979323846f / 180.0f));
    auto put = [](float *d, V3 a) {
        d[0] = a.x;
        d[1] = a.y;
        d[2] = a.z;
    };
    put(out->center, look_from);
    put(out->pixel_00_loc, p00);
    put(out->pixel_delta_u, pdu);
    put(out->pixel_delta_v, pdv);
    put(out->defocus_disk_u, u * defocus_radius);
    put(out->defocus_disk_v, v * defocus_radius);
    out->defocus_angle = s->defocus_angle;
    out->z_near = s->z_near;
    out->z_far = s->z_far;
    return RT_OK;
}

// ------------------------------------------------------------------------------------------------------
// closest hit
// ------------------------------------------------------------------------------------------------------
namespace {
// Closest-hit queries on scenes whose tables exceed this take one lane per ray (measured against the lockstep kernel).
constexpr uint64_t kTraceLanesBytes = 150u * 1024u;
// LDS bytes the extend kernel of a closest-hit query stages the scene tables in (0: read them through the caches).
uint32_t trace_lds_bytes(const rt_context *c, const rt_scene *s) {
    const uint64_t table_bytes = (uint64_t)s->dev.n_nodes * sizeof(rt::Node) + (uint64_t)s->dev.n_prims * sizeof(rt::Prim);
    const bool want_lds = c->pipeline == RT_PIPELINE_AUTO || c->pipeline == RT_PIPELINE_FUSED_LDS ||
                          c->pipeline == RT_PIPELINE_HYBRID_LDS || c->pipeline == RT_PIPELINE_SPLIT_LDS;
    return want_lds && table_bytes <= 40u * 1024u ? (uint32_t)table_bytes : 0u;
}
} // namespace

RT_API int rt_trace_closest_device(rt_context *c, const rt_scene *s, const rt_ray *d_rays, size_t n, float t_min,
                                   float t_max, int use_bvh, rt_hit *d_out, rt_stats *stats) {
    if (!c || !s) return set_error(RT_ERR_INVALID_ARGUMENT, "rt_trace_closest: null context or scene");
    if (s->device != c->device) return set_error(RT_ERR_INVALID_ARGUMENT, "scene and context are on different devices");
    if (n > 0xFFFFFFFFull / 2) return set_error(RT_ERR_INVALID_ARGUMENT, "too many rays in one call");
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n == 0) return RT_OK;
    if (!d_rays || !d_out) return set_error(RT_ERR_INVALID_ARGUMENT, "rt_trace_closest: null ray or hit buffer");
    DeviceGuard guard(c->device);
    if (int rc = ensure_paths(c, n)) return rc;
    const bool count = stats && (c->flags & RT_CTX_COUNTERS) && use_bvh;
    if (count) {
        if (int rc = c->counters.ensure(1)) return rc;
        HIP_TRY(hipMemsetAsync(c->counters.ptr, 0, sizeof(rt::ExtendCounters), c->stream));
    }
    c->spans.clear();
    c->ev_used = 0;
    if (stats) HIP_TRY(hipEventRecord(c->ev_begin, c->stream));
    rt::launch_pack_rays(c->stream, c->max_blocks, reinterpret_cast<const float *>(d_rays), (uint32_t)n, c->ray_o.ptr,
                         c->ray_d.ptr);
    if (int rc = timed_begin(c, K_EXTEND)) return rc;
    const rt::QueueView all{nullptr, nullptr, 0u, (uint32_t)n};
    const uint32_t grid = (uint32_t)std::min<uint64_t>(c->max_blocks, (n + 255) / 256);
    const uint32_t lds_bytes = trace_lds_bytes(c, s);
    // Scenes beyond LDS size: one lane per ray (k_extend_lanes; teapot 5.4 vs 4.0 Grays/s of random rays, 1 M soup 0.58
    // vs 0.30), over the chain records from 2 MB of tables on (queries are incoherent, so the records pay much earlier
    // than in a render: 30 k soup, 3.4 MB, 1.88 vs 1.67 for the walk, 60 k 1.49 vs 1.17, 1 M 0.58 vs 0.30, 10 M 0.24 vs
    // 0.12; the teapot, 1 MB: 3.9 vs 5.4), else over the node table.  RT_LANES_TRAVERSAL forces one of the two for any
    // scene.  The instrumented kernel and the scan of the object list keep the lockstep kernel.
    const uint64_t table_bytes = (uint64_t)s->dev.n_nodes * sizeof(rt::Node) + (uint64_t)s->dev.n_prims * sizeof(rt::Prim);
    const bool lanes = use_bvh && !count && (c->lanes_traversal != 0 || table_bytes > kTraceLanesBytes);
    if (use_bvh && c->trace_traversal &&
        rt::launch_extend_plain(c->stream, grid, s->dev, c->ray_o.ptr, c->ray_d.ptr, c->hit.ptr, (uint32_t)n, t_min, t_max,
                                count ? c->counters.ptr : nullptr, c->trace_traversal == 2)) {
        // the render kernels' walks, on request (a scene whose copy does not fit LDS falls through to the usual kernels)
    } else if (lanes) {
        const bool chain = s->dev.n_prims <= rt::kRefIndex && (c->lanes_traversal > 0 || (c->lanes_traversal == 0 && table_bytes > (2u << 20)));
        if (int rc = chain ? ensure_chain(c, s) : ensure_walk(c, s)) return rc;
        // chain: one 1024-thread block per CU (16 waves: four per SIMD), else six 256-thread blocks
        const uint32_t lanes_grid = chain ? (uint32_t)std::min<uint64_t>(c->cus, (n + 1023) / 1024)
                                          : (uint32_t)std::min<uint64_t>((uint64_t)c->cus * 6u, (n + 255) / 256);
        rt::StackSpill spill{nullptr, lanes_grid * (chain ? 1024u : 256u)};
        if (chain) {
            const uint32_t stack_bytes = (uint32_t)rt::kStackEntriesInLds * 1024u * 8u;
            const uint32_t fit = (160u * 1024u - stack_bytes - 1024u) / (uint32_t)sizeof(rt::ChainNode);
            if (int rc = ensure_chain_top(c, s, c->chain_top_records >= 0 ? std::min<uint32_t>((uint32_t)c->chain_top_records, fit) : fit)) return rc;
        }
        rt::SceneDev dev = s->dev;
        if (chain) {
            // at most two pending entries per interior level (see StackWalk)
            if (int rc = c->stack_spill.ensure((size_t)(2u * s->dev.depth + 2u) * spill.stride)) return rc;
            spill.spill = c->stack_spill.ptr;
        } else {
            dev.top_bytes = top_bytes_for(s, c, 6u);
        }
        rt::launch_extend_lanes(c->stream, lanes_grid, dev, c->ray_o.ptr, c->ray_d.ptr, c->hit.ptr, (uint32_t)n, t_min, t_max,
                                spill, chain);
    } else {
        rt::launch_extend(c->stream, grid, s->dev, c->ray_o.ptr, c->ray_d.ptr, c->hit.ptr, all, t_min, t_max, use_bvh != 0,
                          count ? c->counters.ptr : nullptr, lds_bytes);
    }
    if (int rc = timed_end(c)) return rc;
    rt::launch_finalize_hits(c->stream, c->max_blocks, s->dev, c->ray_o.ptr, c->ray_d.ptr, c->hit.ptr, (uint32_t)n, d_out);
    HIP_TRY(hipGetLastError());
    if (stats) {
        HIP_TRY(hipEventRecord(c->ev_end, c->stream));
        HIP_TRY(hipEventSynchronize(c->ev_end));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_begin, c->ev_end));
        stats->total_ms = ms;
        stats->rays_traced = n;
        stats->extend_launches = 1;
        if (count) {
            rt::ExtendCounters h;
            HIP_TRY(hipMemcpy(&h, c->counters.ptr, sizeof h, hipMemcpyDeviceToHost));
            stats->node_visits = h.node_visits;
            stats->triangle_tests = h.triangle_tests;
            stats->sphere_tests = h.sphere_tests;
        }
        if (int rc = collect_timing(c, stats)) return rc;
    }
    return RT_OK;
}

// The allocation-free variant of rt_trace_closest_device: same kernels, with the ray and hit staging in the
// device's file-scope scratch array instead of the context's workspace.  Nothing here allocates or frees.
RT_API int rt_trace_closest_noalloc(rt_context *c, const rt_scene *s, const rt_ray *d_rays, size_t n, float t_min,
                                    float t_max, int use_bvh, rt_hit *d_out) {
    if (!c || !s) return set_error(RT_ERR_INVALID_ARGUMENT, "rt_trace_closest_noalloc: null context or scene");
    if (s->device != c->device) return set_error(RT_ERR_INVALID_ARGUMENT, "scene and context are on different devices");
    if (n > rt::kNoAllocMaxRays)
        return set_error(RT_ERR_INVALID_ARGUMENT, "rt_trace_closest_noalloc: %zu rays exceed the limit of %u per call", n,
                         rt::kNoAllocMaxRays);
    if (n == 0) return RT_OK;
    if (!d_rays || !d_out) return set_error(RT_ERR_INVALID_ARGUMENT, "rt_trace_closest_noalloc: null ray or hit buffer");
    DeviceGuard guard(c->device);
    if (!guard.ok) return set_error(RT_ERR_DEVICE, "hipSetDevice(%d) failed", c->device);
    float4 *ray_o = c->noalloc_scratch, *ray_d = ray_o + rt::kNoAllocMaxRays, *hits = ray_d + rt::kNoAllocMaxRays;
    rt::launch_pack_rays(c->stream, c->max_blocks, reinterpret_cast<const float *>(d_rays), (uint32_t)n, ray_o, ray_d);
    const rt::QueueView all{nullptr, nullptr, 0u, (uint32_t)n};
    const uint32_t grid = (uint32_t)std::min<uint64_t>(c->max_blocks, (n + 255) / 256);
    rt::launch_extend(c->stream, grid, s->dev, ray_o, ray_d, hits, all, t_min, t_max, use_bvh != 0, nullptr,
                      trace_lds_bytes(c, s));
    rt::launch_finalize_hits(c->stream, c->max_blocks, s->dev, ray_o, ray_d, hits, (uint32_t)n, d_out);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

RT_API int rt_trace_closest(rt_context *c, const rt_scene *s, const rt_ray *rays, size_t n, float t_min, float t_max,
                            int use_bvh, rt_hit *out, rt_stats *stats) {
    if (!c || !s) return set_error(RT_ERR_INVALID_ARGUMENT, "rt_trace_closest: null context or scene");
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n == 0) return RT_OK;
    if (!rays || !out) return set_error(RT_ERR_INVALID_ARGUMENT, "rt_trace_closest: null ray or hit buffer");
    DeviceGuard guard(c->device);
    if (int rc = c->ray_stage.ensure(n * 6)) return rc;
    if (int rc = c->hit_stage.ensure(n * 8)) return rc;
    HIP_TRY(hipMemcpyAsync(c->ray_stage.ptr, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice, c->stream));
    if (int rc = rt_trace_closest_device(c, s, reinterpret_cast<const rt_ray *>(c->ray_stage.ptr), n, t_min, t_max,
                                         use_bvh, reinterpret_cast<rt_hit *>(c->hit_stage.ptr), stats))
        return rc;
    HIP_TRY(hipMemcpyAsync(out, c->hit_stage.ptr, n * sizeof(rt_hit), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

// ------------------------------------------------------------------------------------------------------
// render
// ------------------------------------------------------------------------------------------------------
RT_API int rt_render_device(rt_context *c, const rt_scene *s, const rt_camera *cam, const rt_settings *st,
                            float *d_accum, rt_stats *stats) {
    if (!c || !s || !cam || !st || !d_accum) return set_error(RT_ERR_INVALID_ARGUMENT, "rt_render: null argument");
    if (s->device != c->device) return set_error(RT_ERR_INVALID_ARGUMENT, "scene and context are on different devices");
    if (st->image_width == 0 || st->image_height == 0) return set_error(RT_ERR_INVALID_ARGUMENT, "image size must be non-zero");
    const uint64_t npix64 = (uint64_t)st->image_width * st->image_height;
    if (npix64 > (1ull << 26)) return set_error(RT_ERR_INVALID_ARGUMENT, "image larger than 2^26 pixels");
    if (st->max_depth > 0xFFFFu) return set_error(RT_ERR_INVALID_ARGUMENT, "max_depth larger than 65535");
    if (st->mode > RT_MODE_GAMMA_PER_SAMPLE) return set_error(RT_ERR_INVALID_ARGUMENT, "unknown accumulation mode");
    const uint32_t total = st->sample_count ? st->sample_count : st->samples_per_pixel;
    if ((uint64_t)st->sample_begin + total > (1ull << 22)) return set_error(RT_ERR_INVALID_ARGUMENT, "sample index beyond 2^22");
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (total == 0) return RT_OK;
    const uint32_t npix = (uint32_t)npix64;
    DeviceGuard guard(c->device);

    // samples per batch: as many whole frames as fit in the path budget
    uint64_t per_batch = c->max_paths / npix;
    per_batch = std::max<uint64_t>(1, std::min<uint64_t>(per_batch, std::min<uint32_t>(total, 0xFFFFu)));
    const uint32_t B = (uint32_t)per_batch;
    const size_t slots = (size_t)B * npix;
    if (slots > 0x7FFFFFFFull) return set_error(RT_ERR_INVALID_ARGUMENT, "batch exceeds 2^31 paths");
    const bool count = stats && (c->flags & RT_CTX_COUNTERS);
    // Pipeline choice.  The instrumented extend kernel needs a pipeline that runs extend for every bounce.  Scenes
    // with Mix materials are rendered by the PATHS and SPLIT kernels only (the per-bounce fused kernel has no Mix
    // instance): FUSED / HYBRID requests fall back to SPLIT for them.
    const uint64_t table_bytes = (uint64_t)s->dev.n_nodes * sizeof(rt::Node) + (uint64_t)s->dev.n_prims * sizeof(rt::Prim);
    uint32_t pipeline = c->pipeline;
    // AUTO: tables up to 40 KB go to LDS once per 256-thread block; up to 150 KB once per 1024-thread block, one
    // block per CU (485 spheres, 54 KB: 12.4 vs 8.8 Grays/s from L1/L2); larger scenes are read through the caches.
    // Larger tables: one lane per path (LANES).  Beyond what the L2s hold (32 MiB in all; the crossover is near
    // 48 MB) the tables are read from HBM, where the number of cache lines fetched per ray is what counts: lanes walk
    // the 128-byte chain records from the camera ray on (10 M triangles 0.64 vs 0.37 Grays/s for PATHS, 1 M 1.15 vs
    // 0.74, 600 k 1.30 vs 0.96).  Below that the stackless walk over the 32-byte nodes costs fewer instructions; there
    // bounce 0 runs in lockstep first (neighbouring pixels share their nodes) and lanes take over the survivors
    // (teapot 6.7 vs 4.9 lit and 3.7 vs 2.5 with the default camera, 300 k soup 1.66 vs 1.46; small soups lose a
    // little: 100 k 3.6 vs 3.74, 30 k 7.25 vs 7.5).
    // (the fused kernels' LDS copy stores a node in 48 bytes, see stage_scene_sorted)
    const uint64_t lds_table_bytes = (uint64_t)s->dev.n_nodes * 48u + (uint64_t)s->dev.n_prims * sizeof(rt::Prim);
    if (pipeline == RT_PIPELINE_AUTO) pipeline = lds_table_bytes <= 150u * 1024u ? RT_PIPELINE_PATHS_LDS : RT_PIPELINE_LANES;
    bool lds = pipeline == RT_PIPELINE_FUSED_LDS || pipeline == RT_PIPELINE_HYBRID_LDS || pipeline == RT_PIPELINE_SPLIT_LDS ||
               pipeline == RT_PIPELINE_PATHS_LDS;
    if (count) {
        if (c->pipeline == RT_PIPELINE_AUTO && table_bytes > 64u * 1024u) lds = false;
        pipeline = lds ? RT_PIPELINE_SPLIT_LDS : RT_PIPELINE_SPLIT;
    }
    const bool lanes = pipeline == RT_PIPELINE_LANES;
    const bool lanes_chain = lanes && s->dev.n_prims <= rt::kRefIndex &&
                             (c->lanes_traversal > 0 || (c->lanes_traversal == 0 && table_bytes > (48u << 20)));
    if (lanes)
        if (int rc = lanes_chain ? ensure_chain(c, s) : ensure_walk(c, s)) return rc;
    const bool lanes_lockstep_first = c->lanes_first ? c->lanes_first > 0 : !lanes_chain;
    const int lanes_waves = s->has_mix ? 4 : c->lanes_waves ? c->lanes_waves : (lanes_chain ? 4 : 6);
    if (s->has_mix && pipeline != RT_PIPELINE_PATHS && pipeline != RT_PIPELINE_PATHS_LDS && !lanes)
        pipeline = lds ? RT_PIPELINE_SPLIT_LDS : RT_PIPELINE_SPLIT;
    const bool big_blocks = pipeline == RT_PIPELINE_PATHS_LDS && lds_table_bytes > 40u * 1024u;
    if (lds && lds_table_bytes > (big_blocks ? 150u : 96u) * 1024u)
        return set_error(RT_ERR_INVALID_ARGUMENT, "scene tables (%llu bytes) do not fit the LDS pipeline",
                         (unsigned long long)lds_table_bytes);
    const bool paths = pipeline == RT_PIPELINE_PATHS || pipeline == RT_PIPELINE_PATHS_LDS || lanes;
    const bool fused_all = paths || pipeline == RT_PIPELINE_FUSED || pipeline == RT_PIPELINE_FUSED_LDS;
    const bool fused_first = fused_all || pipeline == RT_PIPELINE_HYBRID || pipeline == RT_PIPELINE_HYBRID_LDS;
    // Traversal of the PATHS kernels: the stack traversal over 64-byte records, or the stackless walk over the
    // 32-byte node table.
    const bool stack = lanes_chain;
    const uint32_t lds_bytes = lds ? (uint32_t)table_bytes : 0u;
    // One grid for every bounce of a batch: block b owns segment b of both queues (see QueueView).
    const uint32_t chunks = (uint32_t)((slots + 255) / 256);
    // A CU holds six of these 256-thread blocks (six waves per SIMD) and whole rounds of six run best.  Scenes
    // whose tables stay in LDS or L1/L2 like many small blocks (four rounds: short tails); scenes read from HBM
    // like one round of large blocks (fewer part-filled waves in the late bounces).  Measured: Cornell 24.6 vs
    // 21.9 Grays/s, 485 spheres 7.9 vs 7.6, teapot 3.65 vs 3.88, 10 M soup 0.217 vs 0.276 (24 vs 6 blocks per CU).
    // Tables beyond what the L2s hold are read from HBM: eight waves per SIMD (and eight blocks per CU) hide more
    // of that latency than six; 10 M soup 0.286 -> 0.325 Grays/s, while the L2-resident teapot loses 2 % with them.
    const bool eight_waves = pipeline == RT_PIPELINE_PATHS && table_bytes > (16u << 20);
    const uint32_t blocks_per_cu = c->blocks_per_cu ? c->blocks_per_cu
                                   : lanes         ? (uint32_t)lanes_waves // one resident round; more chain walks in flight thrash the L2s
                                   : big_blocks    ? 1u
                                   : eight_waves   ? 8u
                                                   : (table_bytes <= 256u * 1024u ? 24u : 6u);
    uint32_t grid = std::min(c->cus * blocks_per_cu, chunks);
    // The lanes kernel over the walk table stages the table's treetop in LDS once per block, so it runs in blocks of
    // `walk_threads` threads, `walk_blocks_per_cu` to a CU, each taking the survivors of grid / walk_grid blocks of the
    // bounce-0 launch (which keeps 256-thread blocks).
    const bool lanes_walk = lanes && !lanes_chain;
    // Measured on the teapot (1 MB of tables; default camera / lit, Grays/s), threads x blocks per CU, treetop bytes:
    // no treetop 3.74 / 6.62; 256 x 6, 24 KB 4.40 / 7.10; 512 x 3, 51 KB 4.63 / 7.37; 768 x 2, 78 KB 4.81 / 7.57; 1024 x 1,
    // 158 KB 4.36 / 7.06 and 512 x 2, 78 KB 4.08 / 6.67 (four waves per SIMD instead of six hide too little latency);
    // eight waves per SIMD (64 registers, spills) 1024 x 2 4.50 / 7.05.
    const uint32_t walk_threads = c->walk_threads ? c->walk_threads : 768u;
    const uint32_t walk_blocks_per_cu = c->walk_blocks_per_cu ? c->walk_blocks_per_cu : std::max(1u, blocks_per_cu * 256u / walk_threads);
    uint32_t walk_grid = grid, lanes_threads = 256u;
    uint32_t top_bytes = 0;
    if (lanes_walk) {
        walk_grid = std::min(c->cus * walk_blocks_per_cu, chunks);
        const uint32_t segs = std::min(32u, std::max(1u, (grid + walk_grid - 1) / walk_grid));
        grid = walk_grid * segs;
        top_bytes = top_bytes_for(s, c, walk_blocks_per_cu);
        lanes_threads = walk_threads;
    } else if (lanes_chain) {
        // The chain walk keeps its stacks in LDS (K entries x 8 bytes per lane); what a CU's LDS has left holds copies of
        // the records nearest the root, which every walk reads, and one block of 1024 threads per CU shares one copy
        // instead of four.  10 M / 1 M soup, Grays/s: no copy 0.729 / 1.300; 256 x 4 blocks, 56 records 0.800 / 1.379; 512 x 2,
        // 120 records 0.823 / 1.428; 1024 x 1, 248 records 0.863 / 1.493; with 8 instead of 16 stack entries in LDS and 184 /
        // 376 / 760 records 0.788 / 1.405, 0.804 / 1.461, 0.842 / 1.561.
        lanes_threads = c->chain_threads ? c->chain_threads : 1024u;
        const uint32_t per_cu = std::max(1u, blocks_per_cu * 256u / lanes_threads);
        walk_grid = std::min(c->cus * per_cu, chunks);
        const uint32_t segs = std::min(32u, std::max(1u, (grid + walk_grid - 1) / walk_grid));
        grid = walk_grid * segs;
        const uint32_t stack_bytes = (uint32_t)rt::kStackEntriesInLds * lanes_threads * 8u;
        const uint32_t budget = (160u * 1024u) / per_cu;
        const uint32_t fit = budget > stack_bytes + 1024u ? (budget - stack_bytes - 1024u) / (uint32_t)sizeof(rt::ChainNode) : 0u;
        const uint32_t want = c->chain_top_records >= 0 ? std::min<uint32_t>((uint32_t)c->chain_top_records, fit) : fit;
        if (int rc = ensure_chain_top(c, s, want)) return rc;
    }
    rt::SceneDev dev = s->dev;
    dev.top_bytes = top_bytes;
    if (lanes_walk && top_bytes == s->dev.n_nodes * (uint32_t)sizeof(rt::WalkNode) &&
        (uint64_t)top_bytes + (uint64_t)s->dev.n_prims * sizeof(rt::Prim) <= (160u * 1024u) / walk_blocks_per_cu - 2048u)
        dev.prims_in_lds = 1u;
    const uint32_t cap = ((chunks + grid - 1) / grid) * 256u;
    if (int rc = ensure_render(c, lanes ? 0 : slots, lanes ? 0 : (size_t)grid * cap, st->max_depth, grid)) return rc;
    if (int rc = c->radiance.ensure(slots)) return rc;
    if (lanes && lanes_lockstep_first) { // survivors of bounce 0 pass through device memory once
        if (int rc = c->ray_o.ensure(slots)) return rc;
        if (int rc = c->ray_d.ensure(slots)) return rc;
        if (int rc = c->beta.ensure(slots)) return rc;
        if (int rc = c->queue[0].ensure((size_t)grid * cap)) return rc;
    }
    const uint32_t block_threads = big_blocks ? 1024u : 256u;
    rt::StackSpill spill{nullptr, std::max(grid * block_threads, walk_grid * lanes_threads)};
    if (stack) { // at most two pending entries per interior level (see StackWalk)
        if (int rc = c->stack_spill.ensure((size_t)(2u * s->dev.depth + 2u) * spill.stride)) return rc;
        spill.spill = c->stack_spill.ptr;
    }
    const uint32_t n_batches = (total + B - 1) / B;
    const size_t counts_len = (size_t)st->max_depth + 1;
    if (stats)
        if (int rc = ensure_host_counts(c, counts_len * n_batches)) return rc;

    rt::RenderParams P{};
    std::memcpy(P.center, cam->center, 12);
    std::memcpy(P.pixel_00_loc, cam->pixel_00_loc, 12);
    std::memcpy(P.pixel_delta_u, cam->pixel_delta_u, 12);
    std::memcpy(P.pixel_delta_v, cam->pixel_delta_v, 12);
    std::memcpy(P.defocus_disk_u, cam->defocus_disk_u, 12);
    std::memcpy(P.defocus_disk_v, cam->defocus_disk_v, 12);
    P.defocus_angle = cam->defocus_angle;
    P.z_near = cam->z_near;
    P.z_far = cam->z_far;
    std::memcpy(P.background, st->background_color, 12);
    P.width = st->image_width;
    P.height = st->image_height;
    P.npix = npix;
    // Camera rays in 8 x 8 pixel tiles where bounce 0 runs in lockstep over tables in the caches (lit teapot 7.62 -> 7.83
    // Grays/s, 100 k soup 4.41 -> 4.85, 300 k soup 2.07 -> 2.31; the teapot seen from inside loses: 4.80 -> 4.67).  Tables in
    // LDS and lanes from the camera ray on gain nothing from it (Cornell 25.2 -> 24.6, 10 M soup 0.655 -> 0.649).
    const bool tiles = lanes && lanes_lockstep_first && st->image_width % 8u == 0u && st->image_height % 8u == 0u;
    P.tiles_x = tiles ? st->image_width / 8u : 0u;
    P.max_depth = st->max_depth;
    P.mode = st->mode;
    P.seed_mix = rt::mix64(st->seed);

    rt::PathState ps{c->ray_o.ptr, c->ray_d.ptr, c->hit.ptr, c->beta.ptr, c->radiance.ptr};
    if (count) HIP_TRY(hipMemsetAsync(c->counters.ptr, 0, sizeof(rt::ExtendCounters), c->stream));
    c->spans.clear();
    c->ev_used = 0;
    if (stats) HIP_TRY(hipEventRecord(c->ev_begin, c->stream));
    uint32_t extend_launches = 0, shade_launches = 0, bounce_launches = 0;

    for (uint32_t bi = 0; bi < n_batches; ++bi) {
        const uint32_t s0 = bi * B;
        const uint32_t b = std::min(B, total - s0);
        const uint32_t n = b * npix;
        P.sample_base = st->sample_begin + s0;
        P.batch_samples = b;
        if (!fused_first) {
            if (int rc = timed_begin(c, K_GENERATE)) return rc;
            rt::launch_generate(c->stream, c->max_blocks, P, ps, n);
            if (int rc = timed_end(c)) return rc;
        }
        if (paths && st->max_depth > 0) { // the whole batch in one launch
            const size_t rows = (size_t)st->max_depth * grid * sizeof(uint32_t);
            HIP_TRY(hipMemsetAsync(c->counts.ptr + grid, 0, rows, c->stream));
            HIP_TRY(hipMemsetAsync(c->early.ptr + grid, 0, rows, c->stream));
            if (int rc = timed_begin(c, K_BOUNCE)) return rc;
            if (lanes)
                rt::launch_paths_lanes(c->stream, grid, dev, P, ps, n, cap, c->queue[0].ptr, c->counts.ptr, c->early.ptr, s->need_uv,
                                       s->has_mix, lanes_waves, spill, lanes_lockstep_first, lanes_chain, lanes_threads, walk_grid);
            else
                rt::launch_paths(c->stream, grid, s->dev, P, ps, n, cap, c->queue[0].ptr, c->queue[1].ptr, c->counts.ptr,
                                 c->early.ptr, s->need_uv, s->has_mix, lds_bytes, big_blocks, eight_waves);
            if (int rc = timed_end(c)) return rc;
            ++bounce_launches;
        }
        for (uint32_t it = 0; it < st->max_depth && !paths; ++it) {
            // bounce `it` consumes queue (it-1)&1 with the per-block counts of row `it`, and produces queue it&1
            // with row it+1; bounce 0 consumes the identity queue.
            rt::QueueView q{it == 0 ? nullptr : c->queue[(it - 1) & 1].ptr, c->counts.ptr + (size_t)it * grid, cap, n};
            uint32_t *q_out = c->queue[it & 1].ptr;
            uint32_t *counts_out = c->counts.ptr + (size_t)(it + 1) * grid;
            uint32_t *early_out = c->early.ptr + (size_t)(it + 1) * grid;
            const bool timed = it < c->timed_bounces;
            if (fused_all || (fused_first && it == 0)) {
                if (int rc = timed_begin(c, it == 0 ? K_BOUNCE0 : K_BOUNCE, timed)) return rc;
                rt::launch_bounce(c->stream, grid, s->dev, P, ps, q, q_out, counts_out, early_out, it == 0, s->need_uv, lds_bytes);
                if (int rc = timed_end(c, timed)) return rc;
                ++bounce_launches;
            } else {
                if (int rc = timed_begin(c, K_EXTEND, timed)) return rc;
                rt::launch_extend(c->stream, grid, s->dev, ps.ray_o, ps.ray_d, ps.hit, q, cam->z_near, cam->z_far, true,
                                  count ? c->counters.ptr : nullptr, lds_bytes);
                if (int rc = timed_end(c, timed)) return rc;
                if (int rc = timed_begin(c, K_SHADE, timed)) return rc;
                rt::launch_shade(c->stream, grid, s->dev, P, ps, q, q_out, counts_out, early_out, s->need_uv, s->has_mix);
                if (int rc = timed_end(c, timed)) return rc;
                ++extend_launches;
                ++shade_launches;
            }
        }
        if (st->max_depth == 0) HIP_TRY(hipMemsetAsync(ps.radiance, 0, (size_t)n * sizeof(float4), c->stream));
        if (int rc = timed_begin(c, K_ACCUMULATE)) return rc;
        rt::launch_accumulate(c->stream, c->max_blocks, P, ps.radiance, d_accum);
        if (int rc = timed_end(c)) return rc;
        if (stats && st->max_depth > 1) {
            rt::launch_sum_counts(c->stream, c->counts.ptr + grid, c->early.ptr + grid, grid, st->max_depth - 1, c->totals.ptr + 1);
            HIP_TRY(hipMemcpyAsync(c->host_counts + counts_len * bi, c->totals.ptr,
                                   counts_len * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        }
    }
    HIP_TRY(hipGetLastError());
    if (stats) {
        HIP_TRY(hipEventRecord(c->ev_end, c->stream));
        HIP_TRY(hipEventSynchronize(c->ev_end));
        HIP_TRY(hipStreamSynchronize(c->stream));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_begin, c->ev_end));
        stats->total_ms = ms;
        uint64_t rays = 0, timed_extend_rays = 0, timed_bounce_rays = 0;
        uint32_t timed_extend = 0, timed_bounce = 0;
        for (uint32_t bi = 0; bi < n_batches; ++bi) {
            const uint32_t b = std::min(B, total - bi * B);
            for (uint32_t it = 0; it < st->max_depth; ++it) {
                // bounce 0 traces every path of the batch
                const uint64_t r = it == 0 ? (uint64_t)b * npix : c->host_counts[counts_len * bi + it];
                rays += r;
                if ((c->flags & RT_CTX_TIMING) && (paths || it < c->timed_bounces)) {
                    if (paths) { // one timed launch per batch covers every bounce
                        timed_bounce_rays += r;
                        if (it == 0) ++timed_bounce;
                    } else if (fused_all || (fused_first && it == 0)) {
                        timed_bounce_rays += r;
                        ++timed_bounce;
                        if (it == 0) {
                            stats->first_bounce_rays += r;
                            ++stats->first_bounce_launches;
                        }
                    } else {
                        timed_extend_rays += r;
                        ++timed_extend;
                    }
                }
            }
        }
        stats->rays_traced = rays;
        stats->timed_extend_rays = timed_extend_rays;
        stats->timed_extend_launches = timed_extend;
        stats->timed_bounce_rays = timed_bounce_rays;
        stats->timed_bounce_launches = timed_bounce;
        stats->bounce_launches = bounce_launches;
        stats->pipeline = pipeline;
        stats->paths = (uint64_t)total * npix;
        stats->extend_launches = extend_launches;
        stats->shade_launches = shade_launches;
        if (count) {
            rt::ExtendCounters h;
            HIP_TRY(hipMemcpy(&h, c->counters.ptr, sizeof h, hipMemcpyDeviceToHost));
            stats->node_visits = h.node_visits;
            stats->triangle_tests = h.triangle_tests;
            stats->sphere_tests = h.sphere_tests;
        }
        if (int rc = collect_timing(c, stats)) return rc;
    }
    return RT_OK;
}

RT_API int rt_render(rt_context *c, const rt_scene *s, const rt_camera *cam, const rt_settings *st, float *accum,
                     rt_stats *stats) {
    if (!c || !s || !cam || !st || !accum) return set_error(RT_ERR_INVALID_ARGUMENT, "rt_render: null argument");
    if (st->image_width == 0 || st->image_height == 0) return set_error(RT_ERR_INVALID_ARGUMENT, "image size must be non-zero");
    const size_t n = (size_t)st->image_width * st->image_height * 3;
    DeviceGuard guard(c->device);
    if (int rc = c->accum.ensure(n)) return rc;
    HIP_TRY(hipMemsetAsync(c->accum.ptr, 0, n * sizeof(float), c->stream));
    if (int rc = rt_render_device(c, s, cam, st, c->accum.ptr, stats)) return rc;
    HIP_TRY(hipMemcpyAsync(accum, c->accum.ptr, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

RT_API int rt_accum_create(rt_context *c, uint32_t width, uint32_t height, float **d_accum) {
    if (!c || !d_accum) return set_error(RT_ERR_INVALID_ARGUMENT, "rt_accum_create: null argument");
    *d_accum = nullptr;
    if (width == 0 || height == 0) return set_error(RT_ERR_INVALID_ARGUMENT, "image size must be non-zero");
    if ((uint64_t)width * height > (1ull << 26)) return set_error(RT_ERR_INVALID_ARGUMENT, "image larger than 2^26 pixels");
    DeviceGuard guard(c->device);
    if (!guard.ok) return set_error(RT_ERR_DEVICE, "hipSetDevice(%d) failed", c->device);
    const size_t bytes = (size_t)width * height * 3 * sizeof(float);
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    if (hipMemsetAsync(p, 0, bytes, c->stream) != hipSuccess) {
        (void)hipFree(p);
        return set_error(RT_ERR_DEVICE, "clearing the accumulator failed");
    }
    *d_accum = static_cast<float *>(p);
    return RT_OK;
}

RT_API int rt_accum_download(rt_context *c, const float *d_accum, size_t count, float *host) {
    if (!c) return set_error(RT_ERR_INVALID_ARGUMENT, "rt_accum_download: null context");
    if (count == 0) return RT_OK;
    if (!d_accum || !host) return set_error(RT_ERR_INVALID_ARGUMENT, "rt_accum_download: null buffer");
    DeviceGuard guard(c->device);
    if (!guard.ok) return set_error(RT_ERR_DEVICE, "hipSetDevice(%d) failed", c->device);
    HIP_TRY(hipMemcpyAsync(host, d_accum, count * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

RT_API int rt_accum_destroy(rt_context *c, float *d_accum) {
    if (!c) return set_error(RT_ERR_INVALID_ARGUMENT, "rt_accum_destroy: null context");
    if (!d_accum) return RT_OK;
    DeviceGuard guard(c->device);
    if (!guard.ok) return set_error(RT_ERR_DEVICE, "hipSetDevice(%d) failed", c->device);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(d_accum));
    return RT_OK;
}

// ------------------------------------------------------------------------------------------------------
// rt_allreduce_sum: RCCL all-reduce over the contexts of one process.  RCCL is bound with dlopen so that the
// library (and single-GPU users) do not depend on it at load time.
// ------------------------------------------------------------------------------------------------------
namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    std::string why;
};

// One communicator set per distinct device list, kept for the life of the process.
struct CommSet {
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
};

std::mutex g_rccl_mutex;
Rccl g_rccl;
std::vector<CommSet> g_comm_sets;

bool load_rccl() {
    if (g_rccl.handle) return true;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names) {
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
        g_rccl.why = dlerror();
    }
    if (!g_rccl.handle) return false;
    bool ok = true;
    auto bind = [&](auto &fn, const char *sym) {
        fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(g_rccl.handle, sym));
        if (!fn) {
            ok = false;
            g_rccl.why = std::string("missing symbol ") + sym;
        }
    };
    bind(g_rccl.comm_init_all, "ncclCommInitAll");
    bind(g_rccl.comm_destroy, "ncclCommDestroy");
    bind(g_rccl.all_reduce, "ncclAllReduce");
    bind(g_rccl.group_start, "ncclGroupStart");
    bind(g_rccl.group_end, "ncclGroupEnd");
    bind(g_rccl.error_string, "ncclGetErrorString");
    if (!ok) {
        dlclose(g_rccl.handle);
        g_rccl.handle = nullptr;
    }
    return ok;
}

} // namespace

RT_API int rt_allreduce_sum(rt_context *const *ctxs, int n_gpus, float *const *d_accum, size_t count) {
    if (!ctxs || !d_accum || n_gpus <= 0) return set_error(RT_ERR_INVALID_ARGUMENT, "rt_allreduce_sum: null argument or no GPUs");
    std::vector<int> devices((size_t)n_gpus);
    for (int i = 0; i < n_gpus; ++i) {
        if (!ctxs[i] || (count && !d_accum[i])) return set_error(RT_ERR_INVALID_ARGUMENT, "rt_allreduce_sum: null entry %d", i);
        devices[(size_t)i] = ctxs[i]->device;
        for (int j = 0; j < i; ++j)
            if (devices[(size_t)j] == devices[(size_t)i])
                return set_error(RT_ERR_INVALID_ARGUMENT, "rt_allreduce_sum: contexts %d and %d share device %d", j, i, devices[(size_t)i]);
    }
    if (count == 0) return RT_OK;
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (!load_rccl()) return set_error(RT_ERR_UNSUPPORTED, "RCCL is not available: %s", g_rccl.why.c_str());
    CommSet *set = nullptr;
    for (auto &s : g_comm_sets)
        if (s.devices == devices) set = &s;
    if (!set) {
        CommSet s;
        s.devices = devices;
        s.comms.resize((size_t)n_gpus);
        ncclResult_t r = g_rccl.comm_init_all(s.comms.data(), n_gpus, devices.data());
        if (r != ncclSuccess) return set_error(RT_ERR_DEVICE, "ncclCommInitAll: %s", g_rccl.error_string(r));
        g_comm_sets.push_back(std::move(s));
        set = &g_comm_sets.back();
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    ncclResult_t r = g_rccl.group_start();
    for (int i = 0; i < n_gpus && r == ncclSuccess; ++i) {
        if (hipSetDevice(devices[(size_t)i]) != hipSuccess) {
            (void)g_rccl.group_end();
            if (prev >= 0) (void)hipSetDevice(prev);
            return set_error(RT_ERR_DEVICE, "hipSetDevice(%d) failed", devices[(size_t)i]);
        }
        r = g_rccl.all_reduce(d_accum[i], d_accum[i], count, ncclFloat, ncclSum, set->comms[(size_t)i], ctxs[i]->stream);
    }
    ncclResult_t e = g_rccl.group_end();
    if (prev >= 0) (void)hipSetDevice(prev);
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) return set_error(RT_ERR_DEVICE, "ncclAllReduce: %s", g_rccl.error_string(r));
    return RT_OK;
}

RT_API int rt_render_node(rt_context *const *ctxs, const rt_scene *const *scenes, int n_gpus, const rt_camera *cam,
                          const rt_settings *st, float *accum) {
    if (!ctxs || !scenes || !cam || !st || !accum || n_gpus <= 0)
        return set_error(RT_ERR_INVALID_ARGUMENT, "rt_render_node: null argument or no GPUs");
    if (st->image_width == 0 || st->image_height == 0) return set_error(RT_ERR_INVALID_ARGUMENT, "image size must be non-zero");
    for (int i = 0; i < n_gpus; ++i) {
        if (!ctxs[i] || !scenes[i]) return set_error(RT_ERR_INVALID_ARGUMENT, "rt_render_node: null entry %d", i);
        if (scenes[i]->device != ctxs[i]->device)
            return set_error(RT_ERR_INVALID_ARGUMENT, "rt_render_node: scene %d lives on device %d, its context on device %d", i,
                             scenes[i]->device, ctxs[i]->device);
    }
    const size_t n = (size_t)st->image_width * st->image_height * 3;
    const uint32_t total = st->sample_count ? st->sample_count : st->samples_per_pixel;
    std::vector<float *> bufs((size_t)n_gpus);
    for (int i = 0; i < n_gpus; ++i) { // enqueue every GPU's share; nothing here waits for a device
        rt_context *c = ctxs[i];
        DeviceGuard guard(c->device);
        if (!guard.ok) return set_error(RT_ERR_DEVICE, "hipSetDevice(%d) failed", c->device);
        if (int rc = c->accum.ensure(n)) return rc;
        bufs[(size_t)i] = c->accum.ptr;
        HIP_TRY(hipMemsetAsync(c->accum.ptr, 0, n * sizeof(float), c->stream));
        rt_settings part = *st;
        rt::sample_block(st->sample_begin, total, (uint32_t)n_gpus, (uint32_t)i, &part.sample_begin, &part.sample_count);
        if (part.sample_count == 0) continue; // more GPUs than samples: this one contributes zeros
        if (int rc = rt_render_device(c, scenes[i], cam, &part, c->accum.ptr, nullptr)) return rc;
    }
    if (n_gpus > 1)
        if (int rc = rt_allreduce_sum(ctxs, n_gpus, bufs.data(), n)) return rc;
    {
        rt_context *c = ctxs[0];
        DeviceGuard guard(c->device);
        HIP_TRY(hipMemcpyAsync(accum, c->accum.ptr, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    for (int i = 0; i < n_gpus; ++i)
        if (int rc = rt_context_synchronize(ctxs[i])) return rc;
    return RT_OK;
}
