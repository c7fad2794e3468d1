/*
 * rt_mi355x.h — C ABI of the MI355X (gfx950) path-tracing core.
 *
 * This is the drop-in boundary for the per-pixel sampling loop of GyrosOfWar/raytracer.  The reference has no
 * FFI of its own; the seam a Rust `-sys` crate would bind is `Renderer::{new, render, render_progressive}`
 * (src/renderer.rs:25,72,105) together with the plain data it is handed: `Camera` (src/camera.rs:6-17),
 * `SceneDescription` (src/scene.rs:18-22), `RenderSettings` (src/scene.rs:53-60), `Material`
 * (src/material.rs:141-148) and `Texture` (src/texture.rs:140-144).  Every entry point below cites the
 * reference interface it stands in for.
 *
 * Conventions
 *   - Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *   - Every call returns an `int`: 0 on success, a negative `rt_status` otherwise; a human-readable message
 *     for the last failure on the calling thread is available from `rt_last_error()`.  Nothing throws or
 *     aborts across the boundary.  Reference panics (`todo!()` in TrowbridgeReitz::scatter,
 *     src/material.rs:213; the infinite recursion of BvhNode::from_objects on an empty object list,
 *     src/bvh.rs:248-285) are reported as error codes from `rt_scene_create`.
 *   - Opaque handles come in create/destroy pairs.  The library copies what it needs from caller buffers and
 *     never frees caller memory.
 *   - A context is bound to one HIP device and one stream.  It is not re-entrant: any host thread may use
 *     it, but not two at once.  Use one context per GPU.
 *   - All arithmetic on the path is IEEE-754 binary32 without FMA contraction, as in the reference
 *     (glam::Vec3A on baseline x86-64).
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------------ */
/* status codes                                                                                           */
/* ------------------------------------------------------------------------------------------------------ */
typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = -1, /* null pointer, index out of range, empty scene, bad enum */
    RT_ERR_UNSUPPORTED = -2,      /* a material the reference itself cannot render (TrowbridgeReitz) */
    RT_ERR_OUT_OF_MEMORY = -3,    /* host or device allocation failed */
    RT_ERR_DEVICE = -4,           /* a HIP call failed; rt_last_error() carries hipGetErrorString */
    RT_ERR_NO_DEVICE = -5,        /* no usable gfx950 device / device index out of range */
    RT_ERR_IO = -6,               /* file could not be read */
    RT_ERR_PARSE = -7             /* malformed glTF / JSON */
} rt_status;

/* ------------------------------------------------------------------------------------------------------ */
/* scene description (host memory, caller-owned)                                                          */
/* ------------------------------------------------------------------------------------------------------ */

/* Texture kinds — `enum Texture` (src/texture.rs:140-144). */
enum { RT_TEX_SOLID = 0, RT_TEX_CHECKER = 1, RT_TEX_IMAGE = 2 };

/* One texture.  SOLID: `rgb` (SolidColor, src/texture.rs:42-51).  CHECKER: `inv_scale`, `even`, `odd` are
 * indices of other textures (Checkerboard, src/texture.rs:53-83); the graph must be acyclic.  IMAGE:
 * `width` x `height` RGB8 texels, row-major, top row first (Image, src/texture.rs:85-136). */
typedef struct rt_texture {
    uint32_t kind;
    float rgb[3];
    float inv_scale;
    int32_t even;
    int32_t odd;
    uint32_t width;
    uint32_t height;
    const uint8_t *pixels;
} rt_texture;

/* Material kinds — `enum Material` (src/material.rs:141-148). */
enum {
    RT_MAT_LAMBERTIAN = 0,
    RT_MAT_METAL = 1,
    RT_MAT_DIELECTRIC = 2,
    RT_MAT_DIFFUSE_LIGHT = 3,
    RT_MAT_MIX = 4,
    RT_MAT_TROWBRIDGE_REITZ = 5 /* rejected with RT_ERR_UNSUPPORTED: the reference panics on it */
};

/* One material.  `texture` indexes `rt_scene_desc.textures` (LAMBERTIAN, METAL, DIFFUSE_LIGHT).  `param` is
 * Metal::fuzz, Dielectric::refraction_index, DiffuseLight::strength, Mix::factor or
 * TrowbridgeReitz::roughness.  `left`/`right` index other materials (MIX only; nesting depth <= 4). */
typedef struct rt_material {
    uint32_t kind;
    int32_t texture;
    float param;
    int32_t left;
    int32_t right;
} rt_material;

/* One triangle mesh — `TriangleMeshData` (src/object/triangle_mesh.rs:12-18) after the loader has applied
 * the node transform to positions and left normals untransformed (src/scene.rs:161-179).  `normals` may hold
 * fewer entries than `positions` (or none): a face any of whose indices has no normal uses the geometric
 * normal (src/object/triangle_mesh.rs:114-124,172-177).  `uvs` is either empty (all hits get (0,0)) or
 * covers every index used (src/object/triangle_mesh.rs:126-136). */
typedef struct rt_mesh {
    uint32_t num_vertices;
    const float *positions; /* 3 * num_vertices */
    uint32_t num_normals;
    const float *normals;   /* 3 * num_normals, may be NULL when num_normals == 0 */
    uint32_t num_uvs;
    const float *uvs;       /* 2 * num_uvs, may be NULL when num_uvs == 0 */
    uint32_t num_faces;
    const uint32_t *indices; /* 3 * num_faces */
    int32_t material;
} rt_mesh;

/* One sphere — `Sphere` (src/object/sphere.rs:11-34). */
typedef struct rt_sphere {
    float center[3];
    float radius;
    int32_t material;
} rt_sphere;

/* The object list of `World` (src/object/world.rs:7-19) before `SceneDescription::build_bvh`
 * (src/scene.rs:29-49).  A primitive's ID is its position in that list: the faces of mesh 0 in index
 * order, then mesh 1, ... (src/scene.rs:238-242), then the spheres — or the spheres first when
 * `spheres_first` is non-zero. */
typedef struct rt_scene_desc {
    uint32_t num_meshes;
    const rt_mesh *meshes;
    uint32_t num_spheres;
    const rt_sphere *spheres;
    uint32_t num_materials;
    const rt_material *materials;
    uint32_t num_textures;
    const rt_texture *textures;
    uint32_t spheres_first;
} rt_scene_desc;

typedef struct rt_scene_info {
    uint32_t num_triangles;
    uint32_t num_spheres;
    uint32_t num_nodes;   /* 2 * primitives - 1 */
    uint32_t depth;       /* number of levels of the BVH, 1 for a single leaf */
    float bounds_min[3];  /* root box, as `scene.root_object.bounding_box()` (src/main.rs:92-95) */
    float bounds_max[3];
    uint64_t device_bytes; /* HBM held by the scene tables */
} rt_scene_info;

/* ------------------------------------------------------------------------------------------------------ */
/* camera and settings                                                                                    */
/* ------------------------------------------------------------------------------------------------------ */

/* `CameraSettings` (src/scene.rs:62-85).  `transform` is a glam Affine3A in column order:
 * x_axis[3], y_axis[3], z_axis[3], translation[3]. */
typedef struct rt_camera_settings {
    float y_fov;
    float z_near;
    float z_far;
    float transform[12];
    float focus_dist;
    float defocus_angle;
} rt_camera_settings;

/* `Camera` (src/camera.rs:6-17), field for field. */
typedef struct rt_camera {
    float center[3];
    float pixel_00_loc[3];
    float pixel_delta_u[3];
    float pixel_delta_v[3];
    float defocus_disk_u[3];
    float defocus_disk_v[3];
    float defocus_angle;
    float z_near;
    float z_far;
} rt_camera;

/* How samples are folded into the accumulator. */
enum {
    RT_MODE_LINEAR_MEAN = 0,     /* Renderer::render (src/renderer.rs:72-103): sum of radiance */
    RT_MODE_GAMMA_PER_SAMPLE = 1 /* Renderer::render_progressive (src/renderer.rs:105-165): sum of
                                    linear_to_gamma(radiance) */
};

/* `RenderSettings` (src/scene.rs:53-60) plus what the reference leaves implicit: the RNG seed (the reference
 * seeds from OS entropy, src/random.rs:9), the accumulation mode, and the sample sub-range this call renders
 * (used to split the sample budget across GPUs; 0 count means [0, samples_per_pixel)). */
typedef struct rt_settings {
    uint32_t image_width;
    uint32_t image_height;
    uint32_t max_depth;
    uint32_t samples_per_pixel;
    float background_color[3];
    uint32_t mode;
    uint64_t seed;
    uint32_t sample_begin;
    uint32_t sample_count;
} rt_settings;

/* ------------------------------------------------------------------------------------------------------ */
/* rays, hits, statistics                                                                                 */
/* ------------------------------------------------------------------------------------------------------ */

/* `Ray` (src/ray.rs:4-7).  The direction is used as given (the reference never normalises camera rays,
 * src/camera.rs:79-81), so `t` is in units of |direction|. */
typedef struct rt_ray {
    float origin[3];
    float direction[3];
} rt_ray;

#define RT_MISS 0xFFFFFFFFu

/* What `Hittable::hit` on the scene root returns (src/object/mod.rs:28-63,67) plus the primitive ID the
 * reference's HitRecord does not carry.  `normal` is face-forwarded as in HitRecord::new and NOT normalised
 * for interpolated mesh normals (src/object/triangle_mesh.rs:172-177).  On a miss prim_id == RT_MISS and
 * the other fields are zero. */
typedef struct rt_hit {
    float t;
    uint32_t prim_id;
    float u;
    float v;
    float normal[3];
    uint32_t front_facing;
} rt_hit;

/* Counters of one render / trace call.  One ray = one call of `world.hit` (src/renderer.rs:44). */
typedef struct rt_stats {
    uint64_t rays_traced;
    uint64_t paths;
    uint64_t node_visits;    /* box tests; only filled when the context has RT_CTX_COUNTERS */
    uint64_t triangle_tests; /* idem */
    uint64_t sphere_tests;   /* idem */
    uint32_t extend_launches;
    uint32_t shade_launches;
    double total_ms;      /* GPU time of the whole call, HIP events on the context's stream */
    double generate_ms;   /* per-kernel sums over the timed launches; need RT_CTX_TIMING */
    double extend_ms;
    double shade_ms;
    double accumulate_ms;
    uint64_t timed_extend_rays;     /* rays processed by the extend launches that extend_ms covers */
    uint32_t timed_extend_launches; /* number of those launches */
    uint32_t pipeline;              /* RT_PIPELINE_* the call actually used */
    double bounce_ms;               /* time of the timed launches of the fused bounce kernel */
    uint64_t timed_bounce_rays;     /* rays processed by the bounce launches that bounce_ms covers */
    uint32_t timed_bounce_launches;
    uint32_t bounce_launches;
    double first_bounce_ms;         /* the part of bounce_ms spent in bounce-0 launches (camera rays) */
    uint64_t first_bounce_rays;     /* rays of those launches */
    uint32_t first_bounce_launches;
    uint32_t reserved;
} rt_stats;

/* ------------------------------------------------------------------------------------------------------ */
/* entry points                                                                                           */
/* ------------------------------------------------------------------------------------------------------ */

typedef struct rt_context rt_context;
typedef struct rt_scene rt_scene;

enum {
    RT_CTX_TIMING = 1u,   /* bracket every kernel launch with HIP events and report per-kernel time */
    RT_CTX_COUNTERS = 2u  /* run the instrumented extend kernel that counts box and primitive tests */
};

/* Replaces `color_eyre::Result` error reporting (src/main.rs:36,68).  Thread-local; never NULL. */
const char *rt_last_error(void);

/* Number of HIP devices visible to the process, or a negative rt_status. */
int rt_device_count(void);

/* Binds a context to HIP device `device`.  `stream` is a `hipStream_t` passed as `void*` (for example
 * torch's current stream); NULL makes the context create and own a stream.  Stands in for the rayon pool the
 * reference renders on (src/renderer.rs:80-82,119-121). */
int rt_context_create(int device, void *stream, uint32_t flags, rt_context **out);
int rt_context_destroy(rt_context *ctx);
/* Blocks until all work queued on the context's stream has finished. */
int rt_context_synchronize(rt_context *ctx);
/* Upper bound on concurrently traced paths (slots of wavefront state, 88 bytes of HBM each).  A render call is
 * cut into batches of at most this many paths; larger batches mean fewer kernel launches.  0 restores the
 * default (2^26). */
int rt_context_set_max_paths(rt_context *ctx, uint64_t max_paths);
/* Which kernels render a batch.  All pipelines produce bit-identical sums.
 *   SPLIT   generate, then per bounce extend + shade (the hit record travels through HBM)
 *   FUSED   per bounce one kernel: (generate +) extend + shade, ray and hit stay in registers
 *   HYBRID  bounce 0 (coherent camera rays) by the fused kernel, later bounces (incoherent rays, very uneven
 *           traversal lengths) by extend with per-lane ray refill, then shade
 *   *_LDS   the same with the BVH and primitive tables staged in LDS (only for scenes whose tables fit)
 *   PATHS   one kernel launch per batch: every block generates its share of camera rays and then keeps
 *           bouncing its own survivors (a block only consumes the queue segment it produced, so no grid-wide
 *           synchronisation is needed between bounces)
 *   LANES   one lane per PATH: a lane keeps its path in registers until it ends, is shaded where it is when its
 *           walk finishes and takes the next path of its block's share when the path ends; no queues.  Tables above
 *           48 MB are read from HBM: the walk runs over 128-byte records holding up to three levels of child boxes
 *           with a per-lane stack in LDS, from the camera ray on, copies of the records nearest the root in LDS.
 *           Smaller tables sit in the L2s: the stackless walk over 32-byte nodes with explicit successors, the top
 *           of the tree staged in LDS by every block, after bounce 0 has been traced with 8 x 8 pixels per wave.
 *   AUTO    PATHS_LDS when the tables fit in 150 KB of LDS (256-thread blocks up to 40 KB, above that one
 *           1024-thread block per CU shares the copy), else LANES: the fastest measured on the benchmark scenes.
 *           The other *_LDS pipelines take tables up to 96 KB.  (Sizes of the LDS copy: 48 bytes per node and per
 *           primitive.) */
enum {
    RT_PIPELINE_AUTO = 0,
    RT_PIPELINE_SPLIT = 1,
    RT_PIPELINE_FUSED = 2,
    RT_PIPELINE_FUSED_LDS = 3,
    RT_PIPELINE_HYBRID = 4,
    RT_PIPELINE_HYBRID_LDS = 5,
    RT_PIPELINE_SPLIT_LDS = 6,
    RT_PIPELINE_PATHS = 7,
    RT_PIPELINE_PATHS_LDS = 8,
    RT_PIPELINE_LANES = 9
};
int rt_context_set_pipeline(rt_context *ctx, uint32_t pipeline);
/* With RT_CTX_TIMING, bracket only the launches of the first `bounces` bounces of every batch with events
 * (the later bounces of a batch carry a vanishing share of the rays but most of the launches).  0 = all. */
int rt_context_set_timed_bounces(rt_context *ctx, uint32_t bounces);

/* `SceneDescription` after `build_bvh(BvhType::Tree)` (src/scene.rs:29-49, src/main.rs:91): validates the
 * description, builds the BVH by the reference's rule (src/bvh.rs:248-285, src/aabb.rs:26-93, 129-137; equal
 * sort keys are ordered by ascending primitive ID), flattens it into the device tables and uploads them.
 * Scenes of at least 65536 primitives (environment variable RT_DEVICE_BUILD_MIN_PRIMS) get the tree built on
 * the GPU, smaller ones on the host; the two builders produce the same tree bit for bit. */
int rt_scene_create(rt_context *ctx, const rt_scene_desc *desc, rt_scene **out);
int rt_scene_destroy(rt_scene *scene);
int rt_scene_get_info(const rt_scene *scene, rt_scene_info *out);

/* Copies the flattened BVH back to the host for inspection: per node in pre-order the box
 * (min xyz, max xyz), the primitive ID of a leaf (or -1) and the index of the node that follows the node's
 * subtree.  Arrays must hold `num_nodes` entries (6 floats per box). */
int rt_scene_export_bvh(const rt_scene *scene, float *boxes, int32_t *leaf_prim, uint32_t *skip);

/* `Camera::new` (src/camera.rs:20-64) and `CameraSettings::default` (src/scene.rs:73-85).  Host only. */
int rt_camera_settings_default(rt_camera_settings *out);
int rt_camera_from_settings(const rt_camera_settings *settings, uint32_t width, uint32_t height,
                            rt_camera *out);

/* `Hittable::hit` on the scene root (src/object/mod.rs:67 -> BvhNode::hit, src/bvh.rs:295-317) for `n`
 * independent rays with the range [t_min, t_max].  Host pointers; the call uploads, traces, downloads and
 * returns when the results are in `out`.  With `use_bvh` == 0 the rays are instead tested against the flat
 * object list in order, as `World::hit` does (src/object/world.rs:21-39). */
int rt_trace_closest(rt_context *ctx, const rt_scene *scene, const rt_ray *rays, size_t n, float t_min,
                     float t_max, int use_bvh, rt_hit *out, rt_stats *stats);

/* Same, with `d_rays` and `d_out` already in device memory of the context's device.  Asynchronous on the
 * context's stream unless `stats` is non-NULL (filling it needs a synchronisation). */
int rt_trace_closest_device(rt_context *ctx, const rt_scene *scene, const rt_ray *d_rays, size_t n,
                            float t_min, float t_max, int use_bvh, rt_hit *d_out, rt_stats *stats);

/* rt_trace_closest_device without any allocation.  No counterpart in the reference: it exists for callers that
 * cannot allocate device memory while they run — code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  Same results as rt_trace_closest_device (no statistics), always by the lockstep closest-hit kernel
 * (for large scenes rt_trace_closest_device runs one lane per ray over a stack whose spill area it sizes to the tree,
 * an allocation); the ray and hit staging that call keeps in the context's workspace lives here in a fixed array in device memory that the library
 * holds on every GPU it runs on until the process ends (RT_NOALLOC_MAX_RAYS x 48 bytes = 3 MiB), whose address
 * was resolved when the context was created.  The call allocates and frees nothing, on the host or the device;
 * `n` > RT_NOALLOC_MAX_RAYS is refused with RT_ERR_INVALID_ARGUMENT.  Asynchronous on the context's stream.  All
 * calls on one GPU share that GPU's array: calls on different streams or from different contexts of the same
 * device that overlap in time race on it; calls on one stream are ordered and safe. */
#define RT_NOALLOC_MAX_RAYS 65536u
int rt_trace_closest_noalloc(rt_context *ctx, const rt_scene *scene, const rt_ray *d_rays, size_t n, float t_min,
                             float t_max, int use_bvh, rt_hit *d_out);

/* `Renderer::render` / passes of `Renderer::render_progressive` (src/renderer.rs:72-156) for the samples
 * [sample_begin, sample_begin + sample_count) of every pixel.  Writes UN-NORMALISED sums over those samples
 * into `accum` (host memory, 3 * width * height floats, row-major RGB, index 3 * (j * width + i) + c as in
 * src/renderer.rs:85-93,124-128); the caller multiplies by 1 / samples as the reference does
 * (src/renderer.rs:75,92,117,140).  The buffer is overwritten. */
int rt_render(rt_context *ctx, const rt_scene *scene, const rt_camera *camera, const rt_settings *settings,
              float *accum, rt_stats *stats);

/* Same with `d_accum` in device memory.  The sums are ADDED to what the buffer holds, sample by sample in
 * ascending sample order, so that successive calls over consecutive sample ranges reproduce the running
 * `aggregate_image` of render_progressive (src/renderer.rs:110,135-137).  Asynchronous on the context's
 * stream unless `stats` is non-NULL. */
int rt_render_device(rt_context *ctx, const rt_scene *scene, const rt_camera *camera,
                     const rt_settings *settings, float *d_accum, rt_stats *stats);

/* The running `aggregate_image` of `Renderer::render_progressive` (src/renderer.rs:110: `vec![0.0; len]`) as a buffer
 * in device memory, for hosts that have no device allocator of their own: 3 * width * height floats, zeroed,
 * on the context's device; pass it to rt_render_device as `d_accum`. */
int rt_accum_create(rt_context *ctx, uint32_t width, uint32_t height, float **d_accum);

/* Copies such a buffer (or any device buffer of `count` floats rt_render_device has written) to `host` after the
 * work enqueued on the context's stream so far, and waits for the copy: what `aggregate_image.iter()` reads when a
 * frame is reported (src/renderer.rs:138-147).  The device buffer keeps its contents. */
int rt_accum_download(rt_context *ctx, const float *d_accum, size_t count, float *host);

/* Frees a buffer of rt_accum_create (after the work enqueued on the context's stream). */
int rt_accum_destroy(rt_context *ctx, float *d_accum);

/* Sums the accumulators of a sample-sharded render across GPUs, in place: afterwards every `d_accum[i]`
 * (device memory of `ctxs[i]`'s device, `count` floats) holds the element-wise sum over all `n_gpus` buffers.
 * This is the one exchange step of `aggregate_image += sample` (src/renderer.rs:135-137) when the samples of a
 * frame are split over the GPUs of one node by `rt_settings.sample_begin/sample_count`; the reference itself is
 * single-process and has no counterpart.  One host process drives all devices: one context per device, all
 * distinct.  The reduction is RCCL's all-reduce (float32, sum) enqueued on each context's stream, so it is
 * ordered after the renders already enqueued there and asynchronous to the host (call
 * `rt_context_synchronize` on each context before reading).  RCCL is loaded at the first call;
 * RT_ERR_UNSUPPORTED if it cannot be.  Multi-process hosts (one process per GPU) use their own communicator
 * instead, e.g. torch.distributed. */
int rt_allreduce_sum(rt_context *const *ctxs, int n_gpus, float *const *d_accum, size_t count);

/* `Renderer::render` / passes of `Renderer::render_progressive` (src/renderer.rs:72-156) over the GPUs of one
 * node, driven by one host process: the samples [sample_begin, sample_begin + sample_count) are split into
 * `n_gpus` contiguous blocks (the first `sample_count % n_gpus` GPUs take one more), GPU i renders its block
 * of every pixel with `scenes[i]` (the same scene created on `ctxs[i]`) into a private accumulator, the
 * accumulators are summed with `rt_allreduce_sum` and the result is copied into `accum` (host memory,
 * 3 * width * height floats, overwritten; un-normalised sums as for `rt_render`).  The GPUs run concurrently.
 * The random stream is keyed on the global sample index, so the set of samples does not depend on `n_gpus`;
 * only the order of the final float additions does.  With `n_gpus` == 1 the result equals `rt_render`'s. */
int rt_render_node(rt_context *const *ctxs, const rt_scene *const *scenes, int n_gpus, const rt_camera *camera,
                   const rt_settings *settings, float *accum);

/* ------------------------------------------------------------------------------------------------------ */
/* glTF import (host only)                                                                                */
/* ------------------------------------------------------------------------------------------------------ */

/* `scene::load_from_gltf` (src/scene.rs:208-250): reads a .gltf (external or base64 buffers) or .glb file the
 * way the reference's loader does — every node's LOCAL transform, the first TRIANGLES primitive of its mesh,
 * positions transformed and normals not, emissive materials as white DiffuseLight of the emissive strength and
 * everything else as Lambertian of the base colour factor, perspective cameras with focus distance 10.
 * Base-colour textures (`load_image`, src/scene.rs:87-114) are decoded as `rt_image_decode` below decodes them.
 * The returned object owns every buffer the scene description points into. */
typedef struct rt_gltf rt_gltf;
int rt_gltf_load(const char *path, rt_gltf **out);
int rt_gltf_free(rt_gltf *gltf);
/* The object list as an rt_scene_desc, valid until rt_gltf_free. */
const rt_scene_desc *rt_gltf_scene(const rt_gltf *gltf);
/* `SceneDescription::cameras` / `SceneDescription::camera(index)` (src/scene.rs:25-27): an index beyond the
 * file's cameras yields CameraSettings::default. */
uint32_t rt_gltf_camera_count(const rt_gltf *gltf);
int rt_gltf_camera(const rt_gltf *gltf, uint32_t index, rt_camera_settings *out);

/* The texels of an image file held in memory as the reference's loader sees them (`gltf::import` decodes the file,
 * `load_image`, src/scene.rs:87-114, wraps the result and `Image::value_at`, src/texture.rs:115-136, reads channels
 * 0..2): `*pixels` receives 3 * width * height bytes, row-major RGB, top row first, (l, l, l) for grey images, alpha
 * dropped; release them with `rt_image_free`.
 *   PNG: 8-bit grey, grey+alpha, RGB, RGBA, palette and 1/2/4-bit grey, interlaced or not; 16-bit samples are refused
 *        as the reference refuses them.  Lossless, so these are the reference's texels.
 *   JPEG: baseline, extended-sequential and progressive Huffman files with 8-bit samples, grey or three components,
 *        chroma subsampled by 1 or 2 per direction; CMYK, arithmetic coding, lossless and 12-bit files are refused.
 *        The integer inverse DCT, the linear chroma upsampling and the integer YCbCr -> RGB conversion follow the
 *        decoder the reference links (zune-jpeg 0.4.11) as far as its algorithms are known; no fixture of the
 *        reference pins these texels.
 * Anything refused or malformed yields RT_ERR_PARSE with the reason in rt_last_error(). */
int rt_image_decode(const void *bytes, size_t size, uint32_t *width, uint32_t *height, uint8_t **pixels);
void rt_image_free(uint8_t *pixels);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */
