/*
 * rt_oracle.c — CPU restatement of the GyrosOfWar/raytracer sampling loop.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker the HIP path is compared with and the timed CPU
 * baseline; it is never linked into, loaded by or called from the product library.  Only tests, the smoke
 * check and the benchmark's cpu_baseline leg may load it.
 *
 * Parity status: the reference (Rust) cannot be built in this environment and its test-suite holds no golden
 * vector for this path, so this restatement is pinned by the reference's source text and by hand-derived
 * known-answer rays on its Cornell asset only.  Third-party arithmetic it depends on (glam 0.28.0 vector ops,
 * rand 0.8.5 float conversion, rayon's unstable sort) is restated from their published behaviour and is
 * "parity unpinned" at the ULP / tie-order level; see each function.
 *
 * Every function cites the reference lines it follows (paths relative to the reference root).
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (binary32, no FMA contraction, SSE scalar math).
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/rt_mi355x.h"

#define ORACLE_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------------------------------------------- */
/* vectors: glam::Vec3A semantics (src/vec3.rs:1-3).                                                      */
/* glam 0.28.0 (not vendored): dot = (x*x' + y*y') + z*z'; cross = product minus product per lane;       */
/* normalize = v * (1 / sqrt(v.v)); Vec3A / f32 is a true division.  No FMA on baseline x86-64.          */
/* ---------------------------------------------------------------------------------------------------- */
typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float vlen2(v3 a) { return vdot(a, a); }
static inline v3 vnormalize(v3 a) { return vscale(a, 1.0f / sqrtf(vdot(a, a))); }
static inline float vat(v3 a, int axis) { return axis == 0 ? a.x : (axis == 1 ? a.y : a.z); }

typedef struct { v3 o, d; } ray_t; /* src/ray.rs:4-17 */
static inline v3 ray_evaluate(const ray_t *r, float t) { return vadd(r->o, vscale(r->d, t)); }

/* src/vec3.rs:12-14 */
static inline v3 reflect(v3 v, v3 n) { return vsub(v, vscale(vscale(n, vdot(v, n)), 2.0f)); }

/* src/vec3.rs:16-21 */
static inline v3 refract(v3 uv, v3 n, float etai_over_etat) {
    float cos_theta = fminf(vdot(vneg(uv), n), 1.0f);
    v3 r_out_perp = vscale(vadd(uv, vscale(n, cos_theta)), etai_over_etat);
    v3 r_out_parallel = vscale(n, -sqrtf(fabsf(1.0f - vlen2(r_out_perp))));
    return vadd(r_out_perp, r_out_parallel);
}

/* ---------------------------------------------------------------------------------------------------- */
/* Range / Aabb (src/range.rs, src/aabb.rs)                                                               */
/* ---------------------------------------------------------------------------------------------------- */
typedef struct { float min, max; } range_t;
typedef struct { range_t x, y, z; } aabb_t;

static const float F_INF = __builtin_inff();

static inline range_t range_new(float mn, float mx) { range_t r = {mn, mx}; return r; }
/* src/range.rs:16-21 */
static inline range_t range_from_ranges(range_t a, range_t b) {
    return range_new(a.min <= b.min ? a.min : b.min, a.max >= b.max ? a.max : b.max);
}
static inline float range_size(range_t r) { return r.max - r.min; }           /* src/range.rs:35-37 */
static inline range_t range_expand(range_t r, float delta) {                  /* src/range.rs:54-60 */
    float padding = delta / 2.0f;
    return range_new(r.min - padding, r.max + padding);
}
static inline int range_contains(range_t r, float v) { return r.min <= v && v <= r.max; }  /* :23-25 */
static inline int range_surrounds(range_t r, float v) { return r.min < v && v < r.max; }   /* :31-33 */
static inline float range_clamp(range_t r, float x) {                         /* src/range.rs:44-52 */
    if (x < r.min) return r.min;
    if (x > r.max) return r.max;
    return x;
}

static inline range_t aabb_interval_at(const aabb_t *b, int axis) {           /* src/aabb.rs:70-76 */
    return axis == 0 ? b->x : (axis == 1 ? b->y : b->z);
}

/* src/aabb.rs:78-93 */
static aabb_t aabb_pad_to_minimums(aabb_t b) {
    const float delta = 0.001f;
    if (range_size(b.x) < delta) b.x = range_expand(b.x, delta);
    if (range_size(b.y) < delta) b.y = range_expand(b.y, delta);
    if (range_size(b.z) < delta) b.z = range_expand(b.z, delta);
    return b;
}

static aabb_t aabb_empty(void) {                                              /* src/aabb.rs:14-18 */
    aabb_t b = {{F_INF, -F_INF}, {F_INF, -F_INF}, {F_INF, -F_INF}};
    return b;
}

/* src/aabb.rs:30-50 */
static aabb_t aabb_from_points(v3 a, v3 b) {
    aabb_t r;
    r.x = a.x <= b.x ? range_new(a.x, b.x) : range_new(b.x, a.x);
    r.y = a.y <= b.y ? range_new(a.y, b.y) : range_new(b.y, a.y);
    r.z = a.z <= b.z ? range_new(a.z, b.z) : range_new(b.z, a.z);
    return aabb_pad_to_minimums(r);
}

/* src/aabb.rs:52-59 */
static aabb_t aabb_from_boxes(aabb_t b0, aabb_t b1) {
    aabb_t r;
    r.x = range_from_ranges(b0.x, b1.x);
    r.y = range_from_ranges(b0.y, b1.y);
    r.z = range_from_ranges(b0.z, b1.z);
    return aabb_pad_to_minimums(r);
}

/* src/aabb.rs:129-137: never compares x with y. */
static int aabb_longest_axis(const aabb_t *b) {
    if (range_size(b->x) > range_size(b->z)) return 0;
    if (range_size(b->y) > range_size(b->z)) return 1;
    return 2;
}

/* src/aabb.rs:95-127: slab test, reciprocal recomputed per call, compare-and-assign (NaN leaves the
 * interval unchanged), reject when max <= min after each axis. */
static int aabb_hit(const aabb_t *b, const ray_t *ray, range_t hr) {
    for (int axis = 0; axis < 3; ++axis) {
        range_t ax = aabb_interval_at(b, axis);
        float ad_inv = 1.0f / vat(ray->d, axis);
        float t0 = (ax.min - vat(ray->o, axis)) * ad_inv;
        float t1 = (ax.max - vat(ray->o, axis)) * ad_inv;
        if (t0 < t1) {
            if (t0 > hr.min) hr.min = t0;
            if (t1 < hr.max) hr.max = t1;
        } else {
            if (t1 > hr.min) hr.min = t1;
            if (t0 < hr.max) hr.max = t0;
        }
        if (hr.max <= hr.min) return 0;
    }
    return 1;
}

/* ---------------------------------------------------------------------------------------------------- */
/* scene tables                                                                                           */
/* ---------------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t nv, nn, nuv, nf;
    v3 *positions;
    v3 *normals;
    float *uvs;
    uint32_t *indices;
    int32_t material;
} mesh_t;

typedef struct {
    uint8_t is_sphere;
    uint32_t mesh;   /* triangle: mesh index */
    uint32_t face;   /* triangle: face index in mesh; sphere: sphere index */
} object_t;

typedef struct {
    aabb_t bbox;
    int32_t left, right; /* interior */
    int32_t object;      /* leaf: object (= primitive) ID, else -1 */
} node_t;

typedef struct { uint32_t kind; v3 rgb; float inv_scale; int32_t even, odd; uint32_t w, h; uint8_t *px; } tex_t;

struct oracle_scene {
    uint32_t n_meshes, n_spheres, n_materials, n_textures, n_objects;
    mesh_t *meshes;
    rt_sphere *spheres;
    aabb_t *sphere_boxes;
    rt_material *materials;
    tex_t *textures;
    object_t *objects;
    node_t *nodes;
    uint32_t n_nodes;
    int32_t root;
    uint32_t depth;
};
typedef struct oracle_scene oracle_scene;

typedef struct { float u, v; } uv_t;

/* src/object/mod.rs:28-36 with the primitive ID added. */
typedef struct {
    v3 point, normal;
    float distance;
    int front_facing;
    int32_t material;
    uv_t uv;
    uint32_t prim;
    float bu, bv; /* barycentrics (triangles) */
} hit_t;

typedef struct { uint64_t nodes, tris, spheres; } counters_t;

/* src/object/mod.rs:39-62 */
static void hit_record_new(hit_t *h, const ray_t *ray, v3 outward_normal, v3 point, float distance,
                           int32_t material, uv_t uv) {
    int front = vdot(ray->d, outward_normal) < 0.0f;
    h->point = point;
    h->normal = front ? outward_normal : vneg(outward_normal);
    h->front_facing = front;
    h->distance = distance;
    h->material = material;
    h->uv = uv;
}

/* src/object/triangle_mesh.rs:104-112 */
static void tri_vertices(const mesh_t *m, uint32_t face, v3 *v0, v3 *v1, v3 *v2) {
    const uint32_t *ix = m->indices + 3 * (size_t)face;
    *v0 = m->positions[ix[0]];
    *v1 = m->positions[ix[1]];
    *v2 = m->positions[ix[2]];
}

/* src/object/triangle_mesh.rs:190-193 */
static aabb_t tri_bounding_box(const mesh_t *m, uint32_t face) {
    v3 v0, v1, v2;
    tri_vertices(m, face, &v0, &v1, &v2);
    return aabb_from_boxes(aabb_from_points(v0, v1), aabb_from_points(v2, v2));
}

/* src/object/triangle_mesh.rs:140-188 (+ :114-136 normals/uv, :204-209 default_normal,
 * src/texture.rs:29-34 tri_lerp). */
static int tri_hit(const mesh_t *m, uint32_t face, const ray_t *ray, range_t hr, hit_t *out) {
    v3 v0, v1, v2;
    tri_vertices(m, face, &v0, &v1, &v2);
    v3 e1 = vsub(v1, v0);
    v3 e2 = vsub(v2, v0);
    v3 p = vcross(ray->d, e2);
    float det = vdot(e1, p);
    if (fabsf(det) < 1.1920929e-07f) return 0; /* f32::EPSILON */
    float inv_det = 1.0f / det;
    v3 tv = vsub(ray->o, v0);
    float u = vdot(tv, p) * inv_det;
    if (u < 0.0f || u > 1.0f) return 0;
    v3 q = vcross(tv, e1);
    float v = vdot(ray->d, q) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return 0;
    float t = vdot(e2, q) * inv_det;
    if (!range_contains(hr, t)) return 0;

    const uint32_t *ix = m->indices + 3 * (size_t)face;
    v3 normal;
    if (ix[0] < m->nn && ix[1] < m->nn && ix[2] < m->nn) {
        v3 n0 = m->normals[ix[0]], n1 = m->normals[ix[1]], n2 = m->normals[ix[2]];
        normal = vadd(vadd(vscale(n0, 1.0f - u - v), vscale(n1, u)), vscale(n2, v));
    } else {
        normal = vnormalize(vcross(e1, e2));
    }
    uv_t uv = {0.0f, 0.0f};
    if (m->nuv != 0) {
        const float *a = m->uvs + 2 * (size_t)ix[0], *b = m->uvs + 2 * (size_t)ix[1],
                    *c = m->uvs + 2 * (size_t)ix[2];
        uv.u = a[0] * (1.0f - u - v) + b[0] * u + c[0] * v;
        uv.v = a[1] * (1.0f - u - v) + b[1] * u + c[1] * v;
    }
    hit_record_new(out, ray, normal, ray_evaluate(ray, t), t, m->material, uv);
    out->bu = u;
    out->bv = v;
    return 1;
}

/* src/object/sphere.rs:42-52 */
static uv_t sphere_get_uv(v3 p) {
    const float PI = 3.14159265358979323846f;
    float theta = acosf(-p.y);
    float phi = atan2f(-p.z, p.x) + PI;
    uv_t r = {phi / (2.0f * PI), theta / PI};
    return r;
}

/* src/object/sphere.rs:55-84 */
static int sphere_hit(const rt_sphere *s, const ray_t *ray, range_t hr, hit_t *out) {
    v3 center = V(s->center[0], s->center[1], s->center[2]);
    v3 oc = vsub(center, ray->o);
    float a = vlen2(ray->d);
    float h = vdot(ray->d, oc);
    float c = vlen2(oc) - s->radius * s->radius;
    float discriminant = h * h - a * c;
    if (discriminant < 0.0f) return 0;
    float sqrtd = sqrtf(discriminant);
    float root = (h - sqrtd) / a;
    if (!range_surrounds(hr, root)) {
        root = (h + sqrtd) / a;
        if (!range_surrounds(hr, root)) return 0;
    }
    v3 point = ray_evaluate(ray, root);
    v3 outward = vdivs(vsub(point, center), s->radius);
    hit_record_new(out, ray, outward, point, root, s->material, sphere_get_uv(point));
    out->bu = 0.0f;
    out->bv = 0.0f;
    return 1;
}

static aabb_t object_bbox(const oracle_scene *sc, uint32_t id) {
    const object_t *o = &sc->objects[id];
    if (o->is_sphere) return sc->sphere_boxes[o->face];
    return tri_bounding_box(&sc->meshes[o->mesh], o->face);
}

static int object_hit(const oracle_scene *sc, uint32_t id, const ray_t *ray, range_t hr, hit_t *out,
                      counters_t *cnt) {
    const object_t *o = &sc->objects[id];
    int r;
    if (o->is_sphere) {
        if (cnt) cnt->spheres++;
        r = sphere_hit(&sc->spheres[o->face], ray, hr, out);
    } else {
        if (cnt) cnt->tris++;
        r = tri_hit(&sc->meshes[o->mesh], o->face, ray, hr, out);
    }
    if (r) out->prim = id;
    return r;
}

/* src/bvh.rs:295-317: box test on every node (leaves included), always left first, the right child sees
 * max = the left hit's distance, `hit_right.or(hit_left)`. */
static int bvh_hit(const oracle_scene *sc, int32_t ni, const ray_t *ray, range_t hr, hit_t *out,
                   counters_t *cnt) {
    const node_t *n = &sc->nodes[ni];
    if (cnt) cnt->nodes++;
    if (!aabb_hit(&n->bbox, ray, hr)) return 0;
    if (n->object >= 0) return object_hit(sc, (uint32_t)n->object, ray, hr, out, cnt);
    hit_t hl, hrr;
    int got_l = bvh_hit(sc, n->left, ray, hr, &hl, cnt);
    range_t r2 = range_new(hr.min, got_l ? hl.distance : hr.max);
    int got_r = bvh_hit(sc, n->right, ray, r2, &hrr, cnt);
    if (got_r) { *out = hrr; return 1; }
    if (got_l) { *out = hl; return 1; }
    return 0;
}

/* src/object/world.rs:21-39: linear scan in object order with a shrinking max. */
static int world_hit(const oracle_scene *sc, const ray_t *ray, range_t hr, hit_t *out, counters_t *cnt) {
    int got = 0;
    float closest = hr.max;
    hit_t h;
    for (uint32_t i = 0; i < sc->n_objects; ++i) {
        if (object_hit(sc, i, ray, range_new(hr.min, closest), &h, cnt)) {
            closest = h.distance;
            *out = h;
            got = 1;
        }
    }
    return got;
}

/* ---------------------------------------------------------------------------------------------------- */
/* BVH build (src/bvh.rs:248-285, src/aabb.rs:61-68)                                                      */
/* The reference sorts with rayon's par_sort_unstable_by_key(Reverse(OrderedFloat(min))) (rayon 1.10.0, not */
/* vendored): the order of equal keys is implementation-defined there.  Here equal keys are ordered by     */
/* ascending primitive ID, which makes the tree a pure function of the object list.                        */
/* ---------------------------------------------------------------------------------------------------- */
typedef struct { float key; uint32_t id; } sort_item;

static int key_is_nan(float f) { return f != f; }

/* Descending by key (OrderedFloat: NaN is the greatest value and equal to itself), then ascending ID. */
static int sort_cmp(const void *pa, const void *pb) {
    const sort_item *a = (const sort_item *)pa, *b = (const sort_item *)pb;
    int an = key_is_nan(a->key), bn = key_is_nan(b->key);
    if (an != bn) return an ? -1 : 1;
    if (!an) {
        if (a->key > b->key) return -1;
        if (a->key < b->key) return 1;
    }
    return (a->id > b->id) - (a->id < b->id);
}

typedef struct {
    oracle_scene *sc;
    aabb_t *obj_boxes;
    uint32_t next_node;
    uint32_t max_depth;
} build_ctx;

static int32_t build_from_objects(build_ctx *bc, uint32_t *ids, uint32_t len, uint32_t level) {
    oracle_scene *sc = bc->sc;
    if (level > bc->max_depth) bc->max_depth = level;
    aabb_t bbox = aabb_empty();
    for (uint32_t i = 0; i < len; ++i) bbox = aabb_from_boxes(bbox, bc->obj_boxes[ids[i]]);
    int axis = aabb_longest_axis(&bbox);
    int32_t me = (int32_t)bc->next_node++;
    node_t *n = &sc->nodes[me];
    if (len == 1) {
        n->bbox = bc->obj_boxes[ids[0]];
        n->object = (int32_t)ids[0];
        n->left = n->right = -1;
        return me;
    }
    sort_item *items = (sort_item *)malloc(sizeof(sort_item) * len);
    for (uint32_t i = 0; i < len; ++i) {
        items[i].key = aabb_interval_at(&bc->obj_boxes[ids[i]], axis).min;
        items[i].id = ids[i];
    }
    qsort(items, len, sizeof(sort_item), sort_cmp);
    for (uint32_t i = 0; i < len; ++i) ids[i] = items[i].id;
    free(items);
    uint32_t mid = len / 2;
    n->bbox = bbox;
    n->object = -1;
    int32_t l = build_from_objects(bc, ids, mid, level + 1);
    int32_t r = build_from_objects(bc, ids + mid, len - mid, level + 1);
    n = &sc->nodes[me];
    n->left = l;
    n->right = r;
    return me;
}

/* ---------------------------------------------------------------------------------------------------- */
/* scene create / destroy                                                                                 */
/* ---------------------------------------------------------------------------------------------------- */
static _Thread_local char g_err[256] = "";
ORACLE_API const char *oracle_last_error(void) { return g_err; }
static int fail(int code, const char *msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

ORACLE_API void oracle_scene_destroy(oracle_scene *sc) {
    if (!sc) return;
    for (uint32_t i = 0; i < sc->n_meshes; ++i) {
        free(sc->meshes[i].positions);
        free(sc->meshes[i].normals);
        free(sc->meshes[i].uvs);
        free(sc->meshes[i].indices);
    }
    for (uint32_t i = 0; i < sc->n_textures; ++i) free(sc->textures[i].px);
    free(sc->meshes);
    free(sc->spheres);
    free(sc->sphere_boxes);
    free(sc->materials);
    free(sc->textures);
    free(sc->objects);
    free(sc->nodes);
    free(sc);
}

static int validate_material(const rt_scene_desc *d, int32_t mi, int depth) {
    if (mi < 0 || (uint32_t)mi >= d->num_materials) return fail(RT_ERR_INVALID_ARGUMENT, "material index out of range");
    const rt_material *m = &d->materials[mi];
    switch (m->kind) {
    case RT_MAT_LAMBERTIAN:
    case RT_MAT_METAL:
    case RT_MAT_DIFFUSE_LIGHT:
        if (m->texture < 0 || (uint32_t)m->texture >= d->num_textures)
            return fail(RT_ERR_INVALID_ARGUMENT, "texture index out of range");
        return 0;
    case RT_MAT_DIELECTRIC:
        return 0;
    case RT_MAT_MIX:
        if (depth >= 4) return fail(RT_ERR_INVALID_ARGUMENT, "Mix nesting deeper than 4");
        if (validate_material(d, m->left, depth + 1)) return RT_ERR_INVALID_ARGUMENT;
        return validate_material(d, m->right, depth + 1);
    case RT_MAT_TROWBRIDGE_REITZ:
        return fail(RT_ERR_UNSUPPORTED, "TrowbridgeReitz::scatter is todo!() in the reference");
    default:
        return fail(RT_ERR_INVALID_ARGUMENT, "unknown material kind");
    }
}

static int validate_texture(const rt_scene_desc *d, int32_t ti, uint32_t depth) {
    if (ti < 0 || (uint32_t)ti >= d->num_textures) return fail(RT_ERR_INVALID_ARGUMENT, "texture index out of range");
    if (depth > d->num_textures) return fail(RT_ERR_INVALID_ARGUMENT, "cyclic checkerboard texture");
    const rt_texture *t = &d->textures[ti];
    if (t->kind == RT_TEX_CHECKER) {
        if (validate_texture(d, t->even, depth + 1)) return RT_ERR_INVALID_ARGUMENT;
        return validate_texture(d, t->odd, depth + 1);
    }
    if (t->kind == RT_TEX_IMAGE && (uint64_t)t->width * t->height != 0 && !t->pixels)
        return fail(RT_ERR_INVALID_ARGUMENT, "image texture without pixels");
    if (t->kind > RT_TEX_IMAGE) return fail(RT_ERR_INVALID_ARGUMENT, "unknown texture kind");
    return 0;
}

ORACLE_API int oracle_scene_create(const rt_scene_desc *d, oracle_scene **out) {
    if (!d || !out) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    *out = NULL;
    uint64_t n_tri = 0;
    for (uint32_t i = 0; i < d->num_meshes; ++i) n_tri += d->meshes[i].num_faces;
    uint64_t n_obj = n_tri + d->num_spheres;
    if (n_obj == 0) return fail(RT_ERR_INVALID_ARGUMENT, "empty scene (BvhNode::from_objects never terminates on it)");
    if (n_obj > 0x7FFFFFFFull) return fail(RT_ERR_INVALID_ARGUMENT, "too many primitives");
    for (uint32_t i = 0; i < d->num_textures; ++i)
        if (validate_texture(d, (int32_t)i, 0)) return RT_ERR_INVALID_ARGUMENT;
    for (uint32_t i = 0; i < d->num_materials; ++i) {
        int rc = validate_material(d, (int32_t)i, 0);
        if (rc) return rc;
    }
    for (uint32_t i = 0; i < d->num_meshes; ++i) {
        const rt_mesh *m = &d->meshes[i];
        if (m->material < 0 || (uint32_t)m->material >= d->num_materials)
            return fail(RT_ERR_INVALID_ARGUMENT, "mesh material out of range");
        for (uint64_t k = 0; k < 3ull * m->num_faces; ++k) {
            if (m->indices[k] >= m->num_vertices) return fail(RT_ERR_INVALID_ARGUMENT, "vertex index out of range");
            if (m->num_uvs != 0 && m->indices[k] >= m->num_uvs) return fail(RT_ERR_INVALID_ARGUMENT, "uv index out of range");
        }
    }
    for (uint32_t i = 0; i < d->num_spheres; ++i)
        if (d->spheres[i].material < 0 || (uint32_t)d->spheres[i].material >= d->num_materials)
            return fail(RT_ERR_INVALID_ARGUMENT, "sphere material out of range");

    oracle_scene *sc = (oracle_scene *)calloc(1, sizeof *sc);
    sc->n_meshes = d->num_meshes;
    sc->n_spheres = d->num_spheres;
    sc->n_materials = d->num_materials;
    sc->n_textures = d->num_textures;
    sc->n_objects = (uint32_t)n_obj;
    sc->meshes = (mesh_t *)calloc(d->num_meshes ? d->num_meshes : 1, sizeof(mesh_t));
    for (uint32_t i = 0; i < d->num_meshes; ++i) {
        const rt_mesh *m = &d->meshes[i];
        mesh_t *o = &sc->meshes[i];
        o->nv = m->num_vertices; o->nn = m->num_normals; o->nuv = m->num_uvs; o->nf = m->num_faces;
        o->material = m->material;
        o->positions = (v3 *)malloc(sizeof(v3) * (o->nv ? o->nv : 1));
        memcpy(o->positions, m->positions, sizeof(float) * 3 * (size_t)o->nv);
        o->normals = (v3 *)malloc(sizeof(v3) * (o->nn ? o->nn : 1));
        if (o->nn) memcpy(o->normals, m->normals, sizeof(float) * 3 * (size_t)o->nn);
        o->uvs = (float *)malloc(sizeof(float) * 2 * (o->nuv ? o->nuv : 1));
        if (o->nuv) memcpy(o->uvs, m->uvs, sizeof(float) * 2 * (size_t)o->nuv);
        o->indices = (uint32_t *)malloc(sizeof(uint32_t) * 3 * (o->nf ? o->nf : 1));
        memcpy(o->indices, m->indices, sizeof(uint32_t) * 3 * (size_t)o->nf);
    }
    sc->spheres = (rt_sphere *)malloc(sizeof(rt_sphere) * (d->num_spheres ? d->num_spheres : 1));
    sc->sphere_boxes = (aabb_t *)malloc(sizeof(aabb_t) * (d->num_spheres ? d->num_spheres : 1));
    for (uint32_t i = 0; i < d->num_spheres; ++i) {
        sc->spheres[i] = d->spheres[i];
        /* src/object/sphere.rs:21-24 */
        float r = d->spheres[i].radius;
        v3 c = V(d->spheres[i].center[0], d->spheres[i].center[1], d->spheres[i].center[2]);
        v3 rv = V(r, r, r);
        sc->sphere_boxes[i] = aabb_from_points(vsub(c, rv), vadd(c, rv));
    }
    sc->materials = (rt_material *)malloc(sizeof(rt_material) * (d->num_materials ? d->num_materials : 1));
    memcpy(sc->materials, d->materials, sizeof(rt_material) * d->num_materials);
    sc->textures = (tex_t *)calloc(d->num_textures ? d->num_textures : 1, sizeof(tex_t));
    for (uint32_t i = 0; i < d->num_textures; ++i) {
        const rt_texture *t = &d->textures[i];
        tex_t *o = &sc->textures[i];
        o->kind = t->kind; o->rgb = V(t->rgb[0], t->rgb[1], t->rgb[2]); o->inv_scale = t->inv_scale;
        o->even = t->even; o->odd = t->odd; o->w = t->width; o->h = t->height;
        size_t bytes = (size_t)t->width * t->height * 3;
        if (t->kind == RT_TEX_IMAGE && bytes) {
            o->px = (uint8_t *)malloc(bytes);
            memcpy(o->px, t->pixels, bytes);
        }
    }
    /* object list: src/scene.rs:238-242 */
    sc->objects = (object_t *)malloc(sizeof(object_t) * n_obj);
    uint32_t k = 0;
    if (d->spheres_first)
        for (uint32_t i = 0; i < d->num_spheres; ++i) { object_t o = {1, 0, i}; sc->objects[k++] = o; }
    for (uint32_t mi = 0; mi < d->num_meshes; ++mi)
        for (uint32_t f = 0; f < d->meshes[mi].num_faces; ++f) { object_t o = {0, mi, f}; sc->objects[k++] = o; }
    if (!d->spheres_first)
        for (uint32_t i = 0; i < d->num_spheres; ++i) { object_t o = {1, 0, i}; sc->objects[k++] = o; }

    sc->n_nodes = 2 * sc->n_objects - 1;
    sc->nodes = (node_t *)malloc(sizeof(node_t) * sc->n_nodes);
    build_ctx bc;
    bc.sc = sc;
    bc.obj_boxes = (aabb_t *)malloc(sizeof(aabb_t) * n_obj);
    bc.next_node = 0;
    bc.max_depth = 0;
    uint32_t *ids = (uint32_t *)malloc(sizeof(uint32_t) * n_obj);
    for (uint32_t i = 0; i < sc->n_objects; ++i) {
        ids[i] = i;
        bc.obj_boxes[i] = object_bbox(sc, i);
    }
    sc->root = build_from_objects(&bc, ids, sc->n_objects, 1);
    sc->depth = bc.max_depth;
    free(ids);
    free(bc.obj_boxes);
    *out = sc;
    return 0;
}

ORACLE_API int oracle_scene_get_info(const oracle_scene *sc, rt_scene_info *out) {
    if (!sc || !out) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    memset(out, 0, sizeof *out);
    out->num_spheres = sc->n_spheres;
    out->num_triangles = sc->n_objects - sc->n_spheres;
    out->num_nodes = sc->n_nodes;
    out->depth = sc->depth;
    const aabb_t *b = &sc->nodes[sc->root].bbox;
    out->bounds_min[0] = b->x.min; out->bounds_min[1] = b->y.min; out->bounds_min[2] = b->z.min;
    out->bounds_max[0] = b->x.max; out->bounds_max[1] = b->y.max; out->bounds_max[2] = b->z.max;
    return 0;
}

/* Pre-order dump of the tree in the layout rt_scene_export_bvh uses. */
static uint32_t export_rec(const oracle_scene *sc, int32_t ni, uint32_t at, float *boxes, int32_t *leaf_prim,
                           uint32_t *skip) {
    const node_t *n = &sc->nodes[ni];
    float *b = boxes + 6 * (size_t)at;
    b[0] = n->bbox.x.min; b[1] = n->bbox.y.min; b[2] = n->bbox.z.min;
    b[3] = n->bbox.x.max; b[4] = n->bbox.y.max; b[5] = n->bbox.z.max;
    leaf_prim[at] = n->object;
    uint32_t next = at + 1;
    if (n->object < 0) {
        next = export_rec(sc, n->left, next, boxes, leaf_prim, skip);
        next = export_rec(sc, n->right, next, boxes, leaf_prim, skip);
    }
    skip[at] = next;
    return next;
}

ORACLE_API int oracle_scene_export_bvh(const oracle_scene *sc, float *boxes, int32_t *leaf_prim, uint32_t *skip) {
    if (!sc || !boxes || !leaf_prim || !skip) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    export_rec(sc, sc->root, 0, boxes, leaf_prim, skip);
    return 0;
}

/* ---------------------------------------------------------------------------------------------------- */
/* random numbers (src/random.rs)                                                                         */
/* The reference draws from a thread-local SmallRng seeded from OS entropy (src/random.rs:8-10), so its   */
/* output is not reproducible.  Two generators are provided behind the same draw functions:               */
/*   COUNTER — the deterministic per-path stream the GPU path uses: draw k of path (pixel, sample) is     */
/*             a pure function of (seed, pixel, sample, k);                                               */
/*   THREAD  — xoshiro256++ per worker thread, the algorithm behind rand 0.8.5's SmallRng on 64-bit       */
/*             targets (not vendored; restated from its published description), seeded from the seed and  */
/*             the worker index.                                                                          */
/* Float conversion follows rand 0.8.5: gen::<f32>() = (next_u32() >> 8) * 2^-24; gen_range(lo..hi) =     */
/* value in [0,1) with 23 random mantissa bits, times (hi - lo), plus lo, redrawn while >= hi.            */
/* ---------------------------------------------------------------------------------------------------- */
enum { ORACLE_RNG_COUNTER = 0, ORACLE_RNG_THREAD = 1 };

typedef struct {
    int mode;
    uint64_t seed_mix; /* COUNTER */
    uint64_t path_key; /* COUNTER: (sample << 42) | (pixel << 16) */
    uint32_t k;        /* COUNTER: index of the next 32-bit draw */
    uint64_t s[4];     /* THREAD */
} rng_t;

static inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

static inline uint32_t rng_next_u32(rng_t *r) {
    if (r->mode == ORACLE_RNG_COUNTER) {
        /* 64-bit block b holds draws 2b (low half) and 2b+1 (high half). */
        uint32_t k = r->k++;
        uint32_t b = (k & 0x0FFFFFFFu) >> 1; /* the stream is defined for 2^28 draws per path */
        uint64_t c = r->path_key | (uint64_t)(b & 0xFFFFu);
        uint64_t bits = mix64(c * 0x9E3779B97F4A7C15ull + r->seed_mix + (uint64_t)(b >> 16) * 0xD6E8FEB86659FD93ull);
        return (k & 1u) ? (uint32_t)(bits >> 32) : (uint32_t)bits;
    }
    uint64_t *s = r->s;
    uint64_t result = rotl64(s[0] + s[3], 23) + s[0];
    uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return (uint32_t)(result >> 32);
}

static void rng_begin_path(rng_t *r, uint32_t pixel, uint32_t sample) {
    if (r->mode == ORACLE_RNG_COUNTER) {
        r->path_key = ((uint64_t)sample << 42) | ((uint64_t)(pixel & 0x3FFFFFFu) << 16);
        r->k = 0;
    }
}

static void rng_seed_thread(rng_t *r, uint64_t seed, uint32_t worker) {
    uint64_t z = seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(worker + 1));
    for (int i = 0; i < 4; ++i) {
        z += 0x9E3779B97F4A7C15ull;
        r->s[i] = mix64(z);
    }
}

/* src/random.rs:12-14 */
static inline float rng_random(rng_t *r) { return (float)(rng_next_u32(r) >> 8) * (1.0f / 16777216.0f); }

/* src/random.rs:16-18 */
static inline float rng_random_range(rng_t *r, float lo, float hi) {
    float scale = hi - lo;
    for (;;) {
        uint32_t bits = (rng_next_u32(r) >> 9) | 0x3F800000u;
        float value1_2;
        memcpy(&value1_2, &bits, 4);
        float value0_1 = value1_2 - 1.0f;
        float res = value0_1 * scale + lo;
        if (res < hi) return res;
    }
}

/* src/vec3.rs:65-72,83-85 */
static v3 gen_unit_vector(rng_t *r) {
    for (;;) {
        float x = rng_random_range(r, -1.0f, 1.0f);
        float y = rng_random_range(r, -1.0f, 1.0f);
        float z = rng_random_range(r, -1.0f, 1.0f);
        v3 p = V(x, y, z);
        if (vlen2(p) < 1.0f) return vnormalize(p);
    }
}

/* src/vec3.rs:74-81 */
static v3 gen_unit_disk(rng_t *r) {
    for (;;) {
        float x = rng_random_range(r, -1.0f, 1.0f);
        float y = rng_random_range(r, -1.0f, 1.0f);
        v3 p = V(x, y, 0.0f);
        if (vlen2(p) < 1.0f) return p;
    }
}

/* ---------------------------------------------------------------------------------------------------- */
/* sampling (src/sample.rs, src/math.rs)                                                                  */
/* ---------------------------------------------------------------------------------------------------- */
#define F_PI_4 0.785398163397448309615660845819875721f
#define F_PI_2 1.57079632679489661923132169163975144f
#define F_1_PI 0.318309886183790671537767526745028724f

/* src/sample.rs:7-23: the tuple is bound as (theta, r) but filled (radius, angle); kept as written. */
static void sample_uniform_disk_concentric(float ux, float uy, float *dx, float *dy) {
    float ox = 2.0f * ux - 1.0f;
    float oy = 2.0f * uy - 1.0f;
    if (ox == 0.0f && oy == 0.0f) { *dx = 1.0f; *dy = 1.0f; return; }
    float theta, r;
    if (fabsf(ox) > fabsf(oy)) {
        theta = ox;
        r = F_PI_4 * (oy / ox);
    } else {
        theta = oy;
        r = F_PI_2 - F_PI_4 * (ox / oy);
    }
    *dx = r * cosf(theta);
    *dy = r * sinf(theta);
}

/* src/sample.rs:25-30, src/math.rs:3-5 */
static v3 cosine_hemisphere(float ux, float uy) {
    float dx, dy;
    sample_uniform_disk_concentric(ux, uy, &dx, &dy);
    float z = sqrtf(fmaxf(1.0f - (dx * dx) - (dy * dy), 0.0f));
    return V(dx, dy, z);
}

/* ---------------------------------------------------------------------------------------------------- */
/* textures (src/texture.rs)                                                                              */
/* ---------------------------------------------------------------------------------------------------- */
static int64_t f32_to_i64_sat(float f) { /* Rust `as i64`: saturating, NaN -> 0 */
    if (f != f) return 0;
    if (f >= 9223372036854775808.0f) return INT64_MAX;
    if (f <= -9223372036854775808.0f) return INT64_MIN;
    return (int64_t)f;
}
static uint32_t f32_to_u32_sat(float f) { /* Rust `as u32` */
    if (!(f > 0.0f)) return 0;
    if (f >= 4294967296.0f) return UINT32_MAX;
    return (uint32_t)f;
}

static v3 texture_value_at(const oracle_scene *sc, int32_t ti, uv_t uv, v3 p) {
    for (;;) {
        const tex_t *t = &sc->textures[ti];
        if (t->kind == RT_TEX_SOLID) return t->rgb;            /* src/texture.rs:47-51 */
        if (t->kind == RT_TEX_CHECKER) {                       /* src/texture.rs:70-83 */
            int64_t x = f32_to_i64_sat(floorf(t->inv_scale * p.x));
            int64_t y = f32_to_i64_sat(floorf(t->inv_scale * p.y));
            int64_t z = f32_to_i64_sat(floorf(t->inv_scale * p.z));
            int64_t sum = (int64_t)((uint64_t)x + (uint64_t)y + (uint64_t)z);
            ti = (sum % 2 == 0) ? t->even : t->odd;
            continue;
        }
        /* src/texture.rs:115-136: the u index is clamped with height - 1 (as written).  An index beyond the
         * row would panic in the reference; it is clamped to the row here. */
        if (t->h == 0 || t->w == 0) return V(0.0f, 1.0f, 1.0f);
        range_t unit = {0.0f, 1.0f};
        float cu = range_clamp(unit, uv.u);
        float cv = 1.0f - range_clamp(unit, uv.v);
        uint32_t i = f32_to_u32_sat(cu * (float)t->w);
        if (i > t->h - 1) i = t->h - 1;
        if (i > t->w - 1) i = t->w - 1;
        uint32_t j = f32_to_u32_sat(cv * (float)t->h);
        if (j > t->h - 1) j = t->h - 1;
        const uint8_t *px = t->px + 3 * ((size_t)j * t->w + i);
        const float color_scale = 1.0f / 255.0f;
        return V((float)px[0] * color_scale, (float)px[1] * color_scale, (float)px[2] * color_scale);
    }
}

/* ---------------------------------------------------------------------------------------------------- */
/* materials (src/material.rs)                                                                            */
/* ---------------------------------------------------------------------------------------------------- */
typedef struct { v3 attenuation; ray_t scattered; int has_pdf; float pdf; } scatter_t;

/* src/material.rs:116-121; powi(5) = x * ((x*x) * (x*x)) (binary exponentiation). */
static float reflectance(float cosine, float refraction_index) {
    float r0 = (1.0f - refraction_index) / (1.0f + refraction_index);
    r0 *= r0;
    float x = 1.0f - cosine;
    float x2 = x * x;
    float x5 = x * (x2 * x2);
    return r0 + (1.0f - r0) * x5;
}

static int material_scatter(const oracle_scene *sc, int32_t mi, const ray_t *ray, const hit_t *hit, rng_t *rng,
                            scatter_t *out) {
    const rt_material *m = &sc->materials[mi];
    switch (m->kind) {
    case RT_MAT_LAMBERTIAN: { /* src/material.rs:47-64: world-space sample, no shading frame */
        v3 w_o = vneg(ray->d);
        float ux = rng_random(rng);
        float uy = rng_random(rng);
        v3 w_i = cosine_hemisphere(ux, uy);
        if (w_o.z < 0.0f) w_i.z *= -1.0f;
        v3 sample = texture_value_at(sc, m->texture, hit->uv, hit->point);
        out->scattered.o = hit->point;
        out->scattered.d = w_i;
        out->attenuation = vscale(sample, F_1_PI);
        out->has_pdf = 1;
        out->pdf = fabsf(w_i.z) * F_1_PI; /* src/sample.rs:32-34, src/math.rs:7-9 */
        return 1;
    }
    case RT_MAT_METAL: { /* src/material.rs:72-82 */
        v3 reflected = reflect(ray->d, hit->normal);
        v3 fuzz = vscale(gen_unit_vector(rng), m->param);
        reflected = vadd(vnormalize(reflected), fuzz);
        out->scattered.o = hit->point;
        out->scattered.d = reflected;
        out->attenuation = texture_value_at(sc, m->texture, hit->uv, hit->point);
        out->has_pdf = 0;
        return 1;
    }
    case RT_MAT_DIELECTRIC: { /* src/material.rs:89-114 */
        float ri = hit->front_facing ? 1.0f / m->param : m->param;
        v3 unit_direction = vnormalize(ray->d);
        float cos_theta = fminf(vdot(vneg(unit_direction), hit->normal), 1.0f);
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        int cannot_refract = ri * sin_theta > 1.0f;
        v3 direction;
        if (cannot_refract || reflectance(cos_theta, ri) > rng_random(rng))
            direction = reflect(unit_direction, hit->normal);
        else
            direction = refract(unit_direction, hit->normal, ri);
        out->attenuation = V(1.0f, 1.0f, 1.0f);
        out->scattered.o = hit->point;
        out->scattered.d = direction;
        out->has_pdf = 0;
        return 1;
    }
    case RT_MAT_DIFFUSE_LIGHT: /* src/material.rs:129-132 */
        return 0;
    case RT_MAT_MIX: { /* src/material.rs:186-204, :21-30; src/random.rs:24-31 */
        if (m->param == 1.0f) return material_scatter(sc, m->right, ray, hit, rng, out);
        if (m->param == 0.0f) return material_scatter(sc, m->left, ray, hit, rng, out);
        scatter_t l, r;
        int gl = material_scatter(sc, m->left, ray, hit, rng, &l);
        int gr = material_scatter(sc, m->right, ray, hit, rng, &r);
        if (gl && gr) {
            float n = rng_random(rng);
            out->scattered = n < m->param ? l.scattered : r.scattered;
            out->attenuation = vadd(vscale(l.attenuation, m->param), vscale(r.attenuation, 1.0f - m->param));
            out->has_pdf = 0;
            return 1;
        }
        if (gr) { *out = r; return 1; }
        if (gl) { *out = l; return 1; }
        return 0;
    }
    default:
        return 0;
    }
}

/* src/material.rs:36-39,134-136: only DiffuseLight overrides emit. */
static v3 material_emit(const oracle_scene *sc, int32_t mi, uv_t uv, v3 p) {
    const rt_material *m = &sc->materials[mi];
    if (m->kind == RT_MAT_DIFFUSE_LIGHT) return vscale(texture_value_at(sc, m->texture, uv, p), m->param);
    return V(0.0f, 0.0f, 0.0f);
}

/* ---------------------------------------------------------------------------------------------------- */
/* camera (src/camera.rs)                                                                                 */
/* ---------------------------------------------------------------------------------------------------- */
static v3 affine_transform_point(const float *m, v3 p) { /* glam Affine3A::transform_point3a */
    v3 xa = V(m[0], m[1], m[2]), ya = V(m[3], m[4], m[5]), za = V(m[6], m[7], m[8]), tr = V(m[9], m[10], m[11]);
    v3 r = vscale(xa, p.x);
    r = vadd(r, vscale(ya, p.y));
    r = vadd(r, vscale(za, p.z));
    return vadd(r, tr);
}
static void put3(float *dst, v3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }
static v3 get3(const float *s) { return V(s[0], s[1], s[2]); }

/* src/scene.rs:73-85 */
ORACLE_API int oracle_camera_settings_default(rt_camera_settings *s) {
    if (!s) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    memset(s, 0, sizeof *s);
    s->y_fov = 80.0f * (3.14159265358979323846f / 180.0f);
    s->z_near = 0.0001f;
    s->z_far = F_INF;
    s->transform[0] = 1.0f; s->transform[4] = 1.0f; s->transform[8] = 1.0f;
    s->focus_dist = 10.0f;
    s->defocus_angle = 0.0f;
    return 0;
}

/* src/camera.rs:20-64 */
ORACLE_API int oracle_camera_from_settings(const rt_camera_settings *s, uint32_t width, uint32_t height, rt_camera *out) {
    if (!s || !out || width == 0 || height == 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad camera arguments");
    v3 v_up = V(0.0f, 1.0f, 0.0f);
    float theta = s->y_fov;
    float h = tanf(theta / 2.0f);
    float viewport_height = 2.0f * h * s->focus_dist;
    float viewport_width = viewport_height * ((float)width / (float)height);
    v3 look_from = affine_transform_point(s->transform, V(0.0f, 0.0f, 0.0f));
    v3 look_at = affine_transform_point(s->transform, V(0.0f, 0.0f, -1.0f));
    v3 w = vnormalize(vsub(look_from, look_at));
    v3 u = vnormalize(vcross(v_up, w));
    v3 v = vcross(w, u);
    v3 center = look_from;
    v3 viewport_u = vscale(u, viewport_width);
    v3 viewport_v = vscale(vneg(v), viewport_height);
    v3 pdu = vdivs(viewport_u, (float)width);
    v3 pdv = vdivs(viewport_v, (float)height);
    v3 upper_left = vsub(vsub(vsub(center, vscale(w, s->focus_dist)), vdivs(viewport_u, 2.0f)), vdivs(viewport_v, 2.0f));
    v3 p00 = vadd(upper_left, vscale(vadd(pdu, pdv), 0.5f));
    float defocus_radius = s->focus_dist * tanf((s->defocus_angle / 2.0f) * (3.14159265358979323846f / 180.0f));
    put3(out->center, center);
    put3(out->pixel_00_loc, p00);
    put3(out->pixel_delta_u, pdu);
    put3(out->pixel_delta_v, pdv);
    put3(out->defocus_disk_u, vscale(u, defocus_radius));
    put3(out->defocus_disk_v, vscale(v, defocus_radius));
    out->defocus_angle = s->defocus_angle;
    out->z_near = s->z_near;
    out->z_far = s->z_far;
    return 0;
}

/* src/camera.rs:68-93 */
static ray_t camera_get_ray(const rt_camera *c, uint32_t i, uint32_t j, rng_t *rng) {
    float ox = rng_random(rng) - 0.5f;
    float oy = rng_random(rng) - 0.5f;
    v3 pixel_sample = vadd(vadd(get3(c->pixel_00_loc), vscale(get3(c->pixel_delta_u), (float)i + ox)),
                           vscale(get3(c->pixel_delta_v), (float)j + oy));
    v3 origin;
    if (c->defocus_angle <= 0.0f) {
        origin = get3(c->center);
    } else {
        v3 p = gen_unit_disk(rng);
        origin = vadd(vadd(get3(c->center), vscale(get3(c->defocus_disk_u), p.x)), vscale(get3(c->defocus_disk_v), p.y));
    }
    ray_t r;
    r.o = origin;
    r.d = vsub(pixel_sample, origin);
    return r;
}

/* ---------------------------------------------------------------------------------------------------- */
/* integrator (src/renderer.rs)                                                                           */
/* ---------------------------------------------------------------------------------------------------- */
typedef struct {
    const oracle_scene *sc;
    const rt_camera *cam;
    const rt_settings *st;
    int use_bvh;
} render_env;

/* src/renderer.rs:33-70 */
static v3 ray_color(const render_env *env, ray_t ray, rng_t *rng, uint64_t *rays) {
    v3 l = V(0.0f, 0.0f, 0.0f);
    v3 beta = V(1.0f, 1.0f, 1.0f);
    uint32_t depth = 0;
    range_t range = range_new(env->cam->z_near, env->cam->z_far);
    while (!(beta.x == 0.0f && beta.y == 0.0f && beta.z == 0.0f)) {
        if (depth >= env->st->max_depth) break;
        hit_t hit;
        int got = env->use_bvh ? bvh_hit(env->sc, env->sc->root, &ray, range, &hit, NULL)
                               : world_hit(env->sc, &ray, range, &hit, NULL);
        ++*rays;
        if (got) {
            l = vadd(l, vmul(beta, material_emit(env->sc, hit.material, hit.uv, hit.point)));
            scatter_t s;
            if (material_scatter(env->sc, hit.material, &ray, &hit, rng, &s)) {
                float pdf = s.has_pdf ? s.pdf : 1.0f;
                v3 f = vdivs(vscale(s.attenuation, fabsf(vdot(s.scattered.d, hit.normal))), pdf);
                beta = vmul(beta, f);
                ray = s.scattered;
            } else {
                break;
            }
        } else {
            l = V(env->st->background_color[0], env->st->background_color[1], env->st->background_color[2]);
            break;
        }
        depth += 1;
    }
    return l;
}

/* src/renderer.rs:159-165 */
static inline float linear_to_gamma(float c) { return c > 0.0f ? sqrtf(c) : 0.0f; }

typedef struct {
    render_env env;
    float *accum;
    atomic_uint_fast64_t next_chunk;
    uint64_t n_chunks;
    uint32_t worker_count;
    int rng_mode;
    uint32_t s_begin, s_count;
    atomic_uint_fast64_t rays;
} job_t;

typedef struct { job_t *job; uint32_t worker; } worker_arg;

/* Pixel loop of Renderer::render (src/renderer.rs:80-98): 256-pixel chunks (square_size 16,
 * src/main.rs:104), all samples of a pixel in ascending order.  In GAMMA_PER_SAMPLE mode each sample goes
 * through linear_to_gamma before it is added, which yields the aggregate_image of render_progressive
 * (src/renderer.rs:115-137) because every pixel's running sum sees its samples in the same order. */
static void *worker_main(void *p) {
    worker_arg *wa = (worker_arg *)p;
    job_t *job = wa->job;
    const rt_settings *st = job->env.st;
    rng_t rng;
    memset(&rng, 0, sizeof rng);
    rng.mode = job->rng_mode;
    rng.seed_mix = mix64(st->seed);
    if (rng.mode == ORACLE_RNG_THREAD) rng_seed_thread(&rng, st->seed, wa->worker);
    uint64_t npix = (uint64_t)st->image_width * st->image_height;
    uint64_t rays = 0;
    for (;;) {
        uint64_t chunk = atomic_fetch_add(&job->next_chunk, 1);
        if (chunk >= job->n_chunks) break;
        uint64_t p0 = chunk * 256, p1 = p0 + 256 < npix ? p0 + 256 : npix;
        for (uint64_t index = p0; index < p1; ++index) {
            uint32_t i = (uint32_t)(index % st->image_width);
            uint32_t j = (uint32_t)(index / st->image_width);
            float *px = job->accum + 3 * index;
            v3 color = V(px[0], px[1], px[2]);
            for (uint32_t s = job->s_begin; s < job->s_begin + job->s_count; ++s) {
                rng_begin_path(&rng, (uint32_t)index, s);
                ray_t ray = camera_get_ray(job->env.cam, i, j, &rng);
                v3 c = ray_color(&job->env, ray, &rng, &rays);
                if (st->mode == RT_MODE_GAMMA_PER_SAMPLE)
                    c = V(linear_to_gamma(c.x), linear_to_gamma(c.y), linear_to_gamma(c.z));
                color = vadd(color, c);
            }
            px[0] = color.x; px[1] = color.y; px[2] = color.z;
        }
    }
    atomic_fetch_add(&job->rays, rays);
    return NULL;
}

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* Adds the un-normalised sums of samples [sample_begin, +sample_count) into `accum` (3*W*H floats). */
ORACLE_API int oracle_render(const oracle_scene *sc, const rt_camera *cam, const rt_settings *st, float *accum,
                             rt_stats *stats, int threads, int rng_mode, int use_bvh) {
    if (!sc || !cam || !st || !accum) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (st->image_width == 0 || st->image_height == 0) return fail(RT_ERR_INVALID_ARGUMENT, "empty image");
    if ((uint64_t)st->image_width * st->image_height > (1ull << 26)) return fail(RT_ERR_INVALID_ARGUMENT, "image too large");
    if (st->mode > RT_MODE_GAMMA_PER_SAMPLE) return fail(RT_ERR_INVALID_ARGUMENT, "bad mode");
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    job_t job;
    job.env.sc = sc; job.env.cam = cam; job.env.st = st; job.env.use_bvh = use_bvh;
    job.accum = accum;
    uint64_t npix = (uint64_t)st->image_width * st->image_height;
    job.n_chunks = (npix + 255) / 256;
    atomic_init(&job.next_chunk, 0);
    atomic_init(&job.rays, 0);
    job.worker_count = (uint32_t)threads;
    job.rng_mode = rng_mode;
    job.s_begin = st->sample_begin;
    job.s_count = st->sample_count ? st->sample_count : st->samples_per_pixel;
    double t0 = now_ms();
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    worker_arg *wa = (worker_arg *)malloc(sizeof(worker_arg) * (size_t)threads);
    for (int i = 0; i < threads; ++i) {
        wa[i].job = &job;
        wa[i].worker = (uint32_t)i;
        pthread_create(&th[i], NULL, worker_main, &wa[i]);
    }
    for (int i = 0; i < threads; ++i) pthread_join(th[i], NULL);
    free(th);
    free(wa);
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->rays_traced = atomic_load(&job.rays);
        stats->paths = npix * job.s_count;
        stats->total_ms = now_ms() - t0;
    }
    return 0;
}

/* `Hittable::hit` on the root for independent rays. */
ORACLE_API int oracle_trace_closest(const oracle_scene *sc, const rt_ray *rays, size_t n, float t_min, float t_max,
                                    int use_bvh, rt_hit *out, rt_stats *stats) {
    if (!sc || (!rays && n) || (!out && n)) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    counters_t cnt = {0, 0, 0};
    for (size_t i = 0; i < n; ++i) {
        ray_t r;
        r.o = get3(rays[i].origin);
        r.d = get3(rays[i].direction);
        hit_t h;
        range_t hr = range_new(t_min, t_max);
        int got = use_bvh ? bvh_hit(sc, sc->root, &r, hr, &h, &cnt) : world_hit(sc, &r, hr, &h, &cnt);
        rt_hit *o = &out[i];
        memset(o, 0, sizeof *o);
        if (got) {
            o->t = h.distance;
            o->prim_id = h.prim;
            o->u = h.bu;
            o->v = h.bv;
            put3(o->normal, h.normal);
            o->front_facing = (uint32_t)h.front_facing;
        } else {
            o->prim_id = RT_MISS;
        }
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->rays_traced = n;
        stats->node_visits = cnt.nodes;
        stats->triangle_tests = cnt.tris;
        stats->sphere_tests = cnt.spheres;
    }
    return 0;
}

/* Exposes the deterministic per-path stream so tests can pin the GPU's copy of it draw by draw. */
ORACLE_API void oracle_rng_draws(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, uint32_t *out) {
    rng_t r;
    memset(&r, 0, sizeof r);
    r.mode = ORACLE_RNG_COUNTER;
    r.seed_mix = mix64(seed);
    rng_begin_path(&r, pixel, sample);
    for (uint32_t i = 0; i < n; ++i) out[i] = rng_next_u32(&r);
}
