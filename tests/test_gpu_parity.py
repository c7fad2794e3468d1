"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same inputs.

Bar: primitive IDs, t, barycentrics and normals bit-exact for closest-hit queries; rendered sums bit-exact
with the oracle running the same per-path random stream (NaN compared as NaN).  The north star only asks for
1e-5 relative on t / normals and an RMSE tolerance on images; bit equality is what the implementation gives.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from helpers import assert_hits_equal, bits_equal, primary_rays, random_rays
from oracle import binding as ob
from raytracer_amd import RenderSettings, SceneDescription, Sphere, _ffi as F
from raytracer_amd.renderer import Context, Renderer, Scene, camera_from_settings
from raytracer_amd.scene import Material, Texture, TriangleMesh
from raytracer_amd.scenes import cornell_with, lit_teapot, rtiow_final_scene, triangle_soup
from test_oracle import check_against_kat, check_bvh_structure, load_kat, mix_scene

pytestmark = pytest.mark.gpu


def both(ctx, desc):
    return Scene(ctx, desc), ob.OracleScene(desc)


def render_pair(ctx, desc, cam_settings, rs, threads=16):
    g, o = both(ctx, desc)
    cam = camera_from_settings(cam_settings, rs.image_width, rs.image_height)
    r = Renderer(cam, g, rs)
    gpu = r.render_sums()
    cpu, cst = o.render(cam, rs, threads=threads)
    return gpu, cpu, r.last_stats, cst


def assert_images_equal(gpu, cpu, min_agree=1.0):
    same = bits_equal(gpu, cpu)
    assert same.mean() >= min_agree, f"{np.count_nonzero(~same)} of {same.size} values differ"


@pytest.mark.parametrize("name", ["cornell", "teapot", "rtiow"])
def test_signed_zero_directions_with_origins_on_box_planes(cornell, teapot, name, monkeypatch):
    """Aabb::hit (src/aabb.rs:95-127) with a direction component of exactly +0 or -0 and the origin exactly on, one
    binary32 below and one above a plane of a node's box: 1 / +0 = +inf, 1 / -0 = -inf and (plane - origin) * inf is a
    NaN on the plane itself, which compare-and-assign treats differently for the two signs (with -0 a box is passed by an
    origin ON its plane, with +0 it is not).  The slab test orders distances with min / max and handles -0 by a
    per-ray substitution; here every box decision is checked through the instrumented kernel's visit counts, which
    must equal the oracle's, next to the hits themselves."""
    desc = {"cornell": cornell, "teapot": teapot}.get(name) or rtiow_final_scene()[0]
    ctx = Context(0, flags=F.RT_CTX_COUNTERS)
    g, o = both(ctx, desc)
    boxes, _, _ = o.export_bvh()
    boxes = np.asarray(boxes, np.float32).reshape(-1, 6)
    rng = np.random.default_rng(77)
    pick = rng.permutation(len(boxes))[:600]
    root_lo, root_hi = boxes[0, :3], boxes[0, 3:]
    rays = []
    for node in pick:
        for axis in range(3):
            for plane in (boxes[node, axis], boxes[node, 3 + axis]):
                for step in (-1, 0, 1):
                    for zero in (np.float32(0.0), np.float32(-0.0)):
                        org = (root_lo + (root_hi - root_lo) * rng.random(3, dtype=np.float32)).astype(np.float32)
                        at = np.float32(plane)
                        if step:
                            at = np.nextafter(at, np.float32(np.inf if step > 0 else -np.inf), dtype=np.float32)
                        org[axis] = at
                        d = rng.normal(size=3).astype(np.float32)
                        d[axis] = zero
                        if rng.random() < 0.25:  # a second zero, of either sign
                            d[(axis + 1) % 3] = np.float32(-0.0) if rng.random() < 0.5 else np.float32(0.0)
                        # aim at the node's box so that the ray meets it and its neighbours
                        centre = (boxes[node, :3] + boxes[node, 3:]) * np.float32(0.5)
                        other = (axis + 2) % 3
                        d[other] = np.float32(centre[other] - org[other]) or np.float32(1.0)
                        rays.append((org, d))
    arr = np.zeros(len(rays), dtype=F.RAY_DTYPE)
    arr["origin"] = np.array([r[0] for r in rays], np.float32)
    arr["direction"] = np.array([r[1] for r in rays], np.float32)
    assert (np.signbit(arr["direction"]) & (arr["direction"] == 0)).sum() > len(arr) // 3  # the -0 components made it through
    for t_min, t_max in ((1e-4, np.inf), (-1.0, 50.0)):
        got, gst = g.trace_closest(arr, t_min, t_max)
        want, ost = o.trace_closest(arr, t_min, t_max)
        assert_hits_equal(got, want)
        assert (gst.node_visits, gst.triangle_tests, gst.sphere_tests) == (ost.node_visits, ost.triangle_tests, ost.sphere_tests)
    # the same rays through the uninstrumented kernels (LDS tables, one lane per ray over the walk table and the chain records)
    plain = Context(0)
    g2 = Scene(plain, desc)
    got, _ = g2.trace_closest(arr, 1e-4, np.inf)
    want, _ = o.trace_closest(arr, 1e-4, np.inf)
    assert_hits_equal(got, want)
    g2.close()
    plain.close()
    g.close()
    ctx.close()
    # ... and through the walks of the fused render kernels over their LDS copies, which order the slabs by the sign of
    # the direction instead (48-byte nodes; eight copies per node for scenes as small as Cornell): hits and visit counts
    for walk in ("sorted", "octants"):
        if walk == "octants" and name != "cornell":
            continue  # eight copies of these trees do not fit LDS
        monkeypatch.setenv("RT_TRACE_TRAVERSAL", walk)
        ctx3 = Context(0, flags=F.RT_CTX_COUNTERS)
        g3 = Scene(ctx3, desc)
        for t_min, t_max in ((1e-4, np.inf), (-1.0, 50.0)):
            got, gst = g3.trace_closest(arr, t_min, t_max)
            want, ost = o.trace_closest(arr, t_min, t_max)
            assert_hits_equal(got, want)
            assert (gst.node_visits, gst.triangle_tests, gst.sphere_tests) == (ost.node_visits, ost.triangle_tests, ost.sphere_tests), walk
        g3.close()
        ctx3.close()
        monkeypatch.delenv("RT_TRACE_TRAVERSAL")


# ---------------------------------------------------------------------------------------------------------
# closest hit
# ---------------------------------------------------------------------------------------------------------
def test_known_answer_rays_gpu(gpu_ctx, cornell):
    kat, rays = load_kat()
    hits, _ = Scene(gpu_ctx, cornell).trace_closest(rays, kat["t_min"], kat["t_max"])
    check_against_kat(kat, hits)


def test_bvh_tables_equal_the_oracles(gpu_ctx, cornell, teapot):
    """The host builder of the product and the oracle's builder are separate code; their trees must be
    identical: boxes bit for bit, same leaf primitives, same skip links."""
    sph, _ = rtiow_final_scene()
    soup, _ = triangle_soup(30000)
    for desc in (cornell, teapot, sph, soup):
        g, o = both(gpu_ctx, desc)
        gb, gp, gs = g.export_bvh()
        obx, op, os_ = o.export_bvh()
        assert np.array_equal(gb.view(np.uint32), obx.view(np.uint32))
        assert np.array_equal(gp, op) and np.array_equal(gs, os_)
        gi, oi = g.info(), o.info()
        assert (gi.num_nodes, gi.depth, gi.num_triangles, gi.num_spheres) == (oi.num_nodes, oi.depth, oi.num_triangles, oi.num_spheres)
        assert list(gi.bounds_min) == list(oi.bounds_min) and list(gi.bounds_max) == list(oi.bounds_max)
    gb, gp, gs = Scene(gpu_ctx, cornell).export_bvh()
    check_bvh_structure(gb, gp, gs, 36)


def test_device_tree_builder_equals_the_host_builder(gpu_ctx, cornell, teapot, monkeypatch):
    """The GPU tree builder (level-by-level radix sorts, csrc/bvh_build.hip) and the host builder give the same
    node table and leaf order, bit for bit, which is also the oracle's: thin axis-aligned walls (the 0.001
    padding fires inside the fold), spheres before and after meshes, duplicated and degenerate triangles, exact
    key ties, one and two objects, and a soup large enough for the block-per-segment levels."""
    from helpers import random_scene

    descs = [cornell, teapot, rtiow_final_scene()[0], triangle_soup(150_000)[0]] + [random_scene(s)[0] for s in range(10)]
    one = SceneDescription()
    one.meshes.append(TriangleMesh(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), np.array([[0, 1, 2]], np.uint32),
                                   np.zeros((0, 3), np.float32), np.zeros((0, 2), np.float32), one.lambertian((0.5, 0.5, 0.5))))
    two = SceneDescription(list(one.meshes), [Sphere((0.0, 0.0, 1.0), 0.5, 0)], list(one.materials), list(one.textures))
    for desc in descs + [one, two]:
        tables = []
        for min_prims in ("1", "4000000000"):
            monkeypatch.setenv("RT_DEVICE_BUILD_MIN_PRIMS", min_prims)
            sc = Scene(gpu_ctx, desc)
            tables.append(sc.export_bvh() + (sc.info().depth,))
        (db, dp, ds, dd), (hb, hp, hs, hd) = tables
        assert np.array_equal(db.view(np.uint32), hb.view(np.uint32)) and np.array_equal(dp, hp) and np.array_equal(ds, hs)
        assert dd == hd
        ob_, op, os_ = ob.OracleScene(desc).export_bvh()
        assert np.array_equal(db.view(np.uint32), ob_.view(np.uint32)) and np.array_equal(dp, op) and np.array_equal(ds, os_)
    # closest hits through a device-built tree, spheres included
    monkeypatch.setenv("RT_DEVICE_BUILD_MIN_PRIMS", "1")
    desc, _ = rtiow_final_scene()
    g, o = both(gpu_ctx, desc)
    rays = random_rays(20000, (-12, 0, -12), (12, 3, 12), seed=2)
    assert_hits_equal(g.trace_closest(rays, 1e-3, np.inf)[0], o.trace_closest(rays, 1e-3, np.inf)[0])


@pytest.mark.parametrize("name", ["cornell", "teapot"])
def test_primary_rays(gpu_ctx, cornell, teapot, name):
    desc = {"cornell": cornell, "teapot": teapot}[name]
    g, o = both(gpu_ctx, desc)
    cs = desc.camera(0)
    cam = camera_from_settings(cs, 320, 240)
    for jitter in (None, 7):
        rays = primary_rays(cam, 320, 240, jitter)
        got, gst = g.trace_closest(rays, cam.z_near, cam.z_far)
        want, _ = o.trace_closest(rays, cam.z_near, cam.z_far)
        assert_hits_equal(got, want)
    assert (got["prim_id"] != F.RT_MISS).mean() > 0.5


def test_lit_teapot_primary_and_shadow_rays(gpu_ctx, teapot):
    """Camera outside the teapot plus 'shadow' rays from hit points towards the light quad (t in (eps, 1))."""
    desc, cs = lit_teapot(teapot)
    g, o = both(gpu_ctx, desc)
    cam = camera_from_settings(cs, 400, 300)
    rays = primary_rays(cam, 400, 300, 3)
    got, _ = g.trace_closest(rays, cam.z_near, cam.z_far)
    want, _ = o.trace_closest(rays, cam.z_near, cam.z_far)
    assert_hits_equal(got, want)
    hit = got["prim_id"] != F.RT_MISS
    assert 0.05 < hit.mean() < 0.9
    p = rays["origin"][hit] + rays["direction"][hit] * got["t"][hit][:, None]
    rng = np.random.default_rng(1)
    target = np.stack([rng.uniform(-15, 15, len(p)), np.full(len(p), 30.0), rng.uniform(-15, 15, len(p))], 1)
    shadow = np.zeros(len(p), dtype=F.RAY_DTYPE)
    shadow["origin"] = p.astype(np.float32)
    shadow["direction"] = (target - p).astype(np.float32)
    got, _ = g.trace_closest(shadow, 1e-3, 1.0)  # t = 1 is the light itself: triangles accept t == t_max
    want, _ = o.trace_closest(shadow, 1e-3, 1.0)
    assert_hits_equal(got, want)
    assert (got["prim_id"] != F.RT_MISS).all() and (got["prim_id"] >= 9128).any()
    got, _ = g.trace_closest(shadow, 1e-3, 0.999)  # occlusion query proper
    want, _ = o.trace_closest(shadow, 1e-3, 0.999)
    assert_hits_equal(got, want)
    assert (got["prim_id"] == F.RT_MISS).any() and (got["prim_id"] != F.RT_MISS).any()


@pytest.mark.parametrize("name", ["cornell", "teapot", "rtiow", "soup"])
def test_random_rays_including_axis_aligned(gpu_ctx, cornell, teapot, name):
    """Incoherent rays with exact zeros in the direction (1/0 = inf, 0*inf = NaN in the slab test)."""
    if name == "rtiow":
        desc, _ = rtiow_final_scene()
        lo, hi = (-12, 0, -12), (12, 3, 12)
    elif name == "soup":
        desc, _ = triangle_soup(200000)
        lo, hi = (-1, -1, -1), (1, 1, 1)
    elif name == "teapot":
        desc, lo, hi = teapot, (-35, -20, -22), (35, 18, 22)
    else:
        desc, lo, hi = cornell, (-1, -1, -1), (1, 1, 1)
    g, o = both(gpu_ctx, desc)
    rays = random_rays(120000, lo, hi, seed=21, axis_aligned_fraction=0.25)
    # origins exactly on box planes of the scene make (bound - origin) == 0 and exercise the NaN path
    rays["origin"][:2000, 2] = np.float32(-1.0)
    rays["origin"][2000:4000, 0] = np.float32(1.0)
    got, _ = g.trace_closest(rays, 1e-4, np.inf)
    want, _ = o.trace_closest(rays, 1e-4, np.inf)
    assert_hits_equal(got, want)
    assert 0.01 < (got["prim_id"] != F.RT_MISS).mean()


def test_world_scan_matches_oracle(gpu_ctx, cornell):
    """use_bvh = 0: World::hit's linear scan in object order (src/object/world.rs:21-39)."""
    g, o = both(gpu_ctx, cornell)
    rays = random_rays(20000, (-1, -1, -1), (1, 1, 3), seed=4)
    kat, krays = load_kat()
    rays = np.concatenate([krays, rays])
    got, _ = g.trace_closest(rays, 0.1, 1000.0, use_bvh=False)
    want, _ = o.trace_closest(rays, 0.1, 1000.0, use_bvh=False)
    assert_hits_equal(got, want)
    assert got["prim_id"][0] == 1  # the later object wins the tie on the shared diagonal
    sph, _ = rtiow_final_scene()
    g, o = both(gpu_ctx, sph)
    rays = random_rays(5000, (-12, 0, -12), (12, 3, 12), seed=8)
    got, _ = g.trace_closest(rays, 1e-3, np.inf, use_bvh=False)
    want, _ = o.trace_closest(rays, 1e-3, np.inf, use_bvh=False)
    assert_hits_equal(got, want)


def test_trace_edge_cases(gpu_ctx, cornell):
    g, o = both(gpu_ctx, cornell)
    empty = np.zeros(0, dtype=F.RAY_DTYPE)
    hits, st = g.trace_closest(empty, 0.1, 10.0)
    assert len(hits) == 0
    # one ray; NaN and inf components; zero direction
    rays = np.zeros(5, dtype=F.RAY_DTYPE)
    rays["origin"] = (0, 0, 3.78)
    rays["direction"][0] = (0, 0, -1)
    rays["direction"][1] = (np.nan, 0, -1)
    rays["direction"][2] = (0, 0, 0)
    rays["direction"][3] = (np.inf, 0, -1)
    rays["direction"][4] = (1e-30, 1e-30, -1e-30)
    got, _ = g.trace_closest(rays, 0.1, 1000.0)
    want, _ = o.trace_closest(rays, 0.1, 1000.0)
    assert_hits_equal(got, want)
    # instrumented counters agree with the oracle's visit counts
    ctx2 = Context(0, flags=F.RT_CTX_COUNTERS | F.RT_CTX_TIMING)
    g2 = Scene(ctx2, cornell)
    rays = random_rays(50000, (-1, -1, -1), (1, 1, 3), seed=2)
    got, gst = g2.trace_closest(rays, 0.1, 1000.0)
    want, ost = o.trace_closest(rays, 0.1, 1000.0)
    assert (gst.node_visits, gst.triangle_tests, gst.sphere_tests) == (ost.node_visits, ost.triangle_tests, ost.sphere_tests)
    assert gst.extend_ms > 0 and gst.total_ms >= gst.extend_ms
    g2.close()
    ctx2.close()


# ---------------------------------------------------------------------------------------------------------
# rendering
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", [F.RT_MODE_LINEAR_MEAN, F.RT_MODE_GAMMA_PER_SAMPLE])
def test_cornell_render_bit_exact(gpu_ctx, cornell, mode):
    rs = RenderSettings(200, 150, 0, 50, 12, mode=mode, seed=1)
    gpu, cpu, gst, cst = render_pair(gpu_ctx, cornell, cornell.camera(0), rs)
    assert_images_equal(gpu, cpu)
    assert gst.rays_traced == cst.rays_traced and gst.paths == cst.paths == 200 * 150 * 12
    assert np.isinf(gpu).any()
    if mode == F.RT_MODE_GAMMA_PER_SAMPLE:
        assert not np.isnan(gpu).any()


def test_config1_cornell_600(gpu_ctx, cornell):
    """The reference's own CPU-runnable case: 600 x 600, depth 50, progressive (gamma per sample); 10 of its
    100 samples are enough for a bit-exact comparison."""
    rs = RenderSettings(600, 600, 0, 50, 100, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=1, sample_begin=40, sample_count=10)
    gpu, cpu, gst, cst = render_pair(gpu_ctx, cornell, cornell.camera(0), rs)
    assert_images_equal(gpu, cpu)
    assert gst.rays_traced == cst.rays_traced


def test_materials_metal_dielectric_checker(gpu_ctx, cornell):
    desc = cornell_with(cornell)
    rs = RenderSettings(160, 160, 0, 50, 8, mode=F.RT_MODE_LINEAR_MEAN, seed=5, background_color=(0.1, 0.2, 0.3))
    gpu, cpu, gst, cst = render_pair(gpu_ctx, desc, desc.camera(0), rs)
    assert_images_equal(gpu, cpu)
    assert gst.rays_traced == cst.rays_traced


def test_rtiow_spheres_with_defocus_and_background(gpu_ctx):
    """Config 3 at reduced size: spheres, all three RTIOW materials, thin-lens camera (rejection-sampled disk),
    constant background.  Bit-exact because sphere UVs (the only acos/atan2 users) are not sampled here."""
    desc, cs = rtiow_final_scene()
    rs = RenderSettings(240, 135, 0, 50, 6, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=2, background_color=(0.7, 0.8, 1.0))
    gpu, cpu, gst, cst = render_pair(gpu_ctx, desc, cs, rs)
    assert_images_equal(gpu, cpu)
    assert gst.rays_traced == cst.rays_traced
    assert gst.rays_traced / gst.paths > 1.5


LDS_PIPELINES = (F.RT_PIPELINE_FUSED_LDS, F.RT_PIPELINE_HYBRID_LDS, F.RT_PIPELINE_SPLIT_LDS, F.RT_PIPELINE_PATHS_LDS)


@pytest.mark.parametrize("pipeline", [F.RT_PIPELINE_SPLIT, F.RT_PIPELINE_FUSED, F.RT_PIPELINE_FUSED_LDS,
                                      F.RT_PIPELINE_HYBRID, F.RT_PIPELINE_HYBRID_LDS, F.RT_PIPELINE_SPLIT_LDS,
                                      F.RT_PIPELINE_PATHS, F.RT_PIPELINE_PATHS_LDS, F.RT_PIPELINE_LANES])
def test_every_pipeline_matches_the_oracle(cornell, teapot, pipeline):
    """Split kernels (generate / extend / shade), the fused per-bounce kernel, the hybrid of the two and the
    one-launch-per-batch kernel, each with the scene tables in HBM or staged in LDS, must all give the oracle's
    sums bit for bit."""
    ctx = Context(0, flags=F.RT_CTX_TIMING)
    ctx.set_pipeline(pipeline)
    cases = [(cornell_with(cornell), None, RenderSettings(120, 90, 0, 50, 6, mode=F.RT_MODE_LINEAR_MEAN, seed=3,
                                                          background_color=(0.2, 0.1, 0.3))),
             (cornell, None, RenderSettings(150, 100, 0, 50, 9, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=4)),
             (cornell, None, RenderSettings(31, 17, 0, 1, 3, mode=F.RT_MODE_LINEAR_MEAN, seed=5)),
             (cornell, None, RenderSettings(9, 9, 0, 0, 2, mode=F.RT_MODE_LINEAR_MEAN, seed=5))]
    sph, sph_cam = rtiow_final_scene()
    cases.append((sph, sph_cam, RenderSettings(160, 90, 0, 50, 4, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=6,
                                               background_color=(0.7, 0.8, 1.0))))
    if pipeline not in LDS_PIPELINES:
        lit, lit_cam = lit_teapot(teapot)
        cases.append((lit, lit_cam, RenderSettings(120, 80, 0, 50, 3, mode=F.RT_MODE_LINEAR_MEAN, seed=7)))
    for desc, cs, rs in cases:
        gpu, cpu, gst, cst = render_pair(ctx, desc, cs or desc.camera(0), rs)
        assert_images_equal(gpu, cpu)
        assert gst.rays_traced == cst.rays_traced
        assert gst.pipeline == pipeline
    if pipeline in LDS_PIPELINES:  # tables too large for LDS -> a clear error, not a fault
        with pytest.raises(F.RtError) as e:
            render_pair(ctx, teapot, teapot.camera(0), RenderSettings(16, 16, 0, 4, 1))
        assert e.value.code == F.RT_ERR_INVALID_ARGUMENT and "LDS" in e.value.message
    ctx.close()


@pytest.mark.parametrize("traversal", ["walk", "chain"])
def test_closest_hit_queries_one_lane_per_ray(cornell, teapot, monkeypatch, traversal):
    """rt_trace_closest on scenes beyond LDS size runs one lane per ray (k_extend_lanes), over the node table or over
    the chain records depending on the scene's size; here either traversal is forced for every scene, including
    LDS-sized ones, the degenerate trees of one and two primitives, and ray counts that leave waves part-filled:
    hits equal the oracle's bit for bit."""
    monkeypatch.setenv("RT_LANES_TRAVERSAL", traversal)
    ctx = Context(0)
    sph, _ = rtiow_final_scene()
    soup, _ = triangle_soup(60000)
    one = SceneDescription()
    one.meshes.append(TriangleMesh(np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], np.float32), np.array([[0, 1, 2]], np.uint32),
                                   np.zeros((0, 3), np.float32), np.zeros((0, 2), np.float32), one.lambertian((0.5, 0.6, 0.7))))
    two = SceneDescription(list(one.meshes), [Sphere((0.0, 0.0, -1.0), 0.5, 0)], list(one.materials), list(one.textures))
    cases = [(cornell, (-1, -1, -1), (1, 1, 1), 70001), (teapot, (-35, -20, -22), (35, 18, 22), 50000),
             (sph, (-12, 0, -12), (12, 3, 12), 30000), (soup, (-1, -1, -1), (1, 1, 1), 100000),
             (one, (-1, -1, -2), (1, 1, 2), 1), (one, (-1, -1, -2), (1, 1, 2), 77), (two, (-1, -1, -2), (1, 1, 2), 4099)]
    for k, (desc, lo, hi, n) in enumerate(cases):
        g, o = both(ctx, desc)
        rays = random_rays(n, lo, hi, seed=40 + k, axis_aligned_fraction=0.25)
        rays["origin"][: n // 50, 2] = np.float32(lo[2])
        for t_min, t_max in ((1e-4, np.inf), (0.05, 1.5)):
            got, _ = g.trace_closest(rays, t_min, t_max)
            want, _ = o.trace_closest(rays, t_min, t_max)
            assert_hits_equal(got, want)
    ctx.close()


@pytest.mark.parametrize("traversal", ["walk", "chain"])
@pytest.mark.parametrize("first", ["lockstep", "lanes"])
def test_lanes_pipeline_variants(cornell, teapot, monkeypatch, traversal, first):
    """The one-lane-per-path pipeline with either traversal (stackless walk over the 32-byte nodes, per-lane stack
    over the 128-byte chain records) and either way of tracing bounce 0 (per-bounce kernel in lockstep first, or
    lanes from the camera ray on), whatever AUTO would pick for the scene: sums and ray counts equal the oracle's.
    Scenes: every material kind, spheres with defocus, a mesh deep enough for the stack to spill out of LDS, a
    Mix scene, and the degenerate trees of one and two primitives (the root is a leaf / the only record)."""
    monkeypatch.setenv("RT_LANES_TRAVERSAL", traversal)
    monkeypatch.setenv("RT_LANES_FIRST", first)
    ctx = Context(0)
    ctx.set_pipeline(F.RT_PIPELINE_LANES)
    sph, sph_cam = rtiow_final_scene()
    lit, lit_cam = lit_teapot(teapot)
    soup, soup_cam = triangle_soup(60000, emissive_fraction=0.02)
    one = SceneDescription()
    one.meshes.append(TriangleMesh(np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], np.float32), np.array([[0, 1, 2]], np.uint32),
                                   np.zeros((0, 3), np.float32), np.zeros((0, 2), np.float32), one.lambertian((0.5, 0.6, 0.7))))
    two = SceneDescription(list(one.meshes), [Sphere((0.0, 0.0, -1.0), 0.5, 0)], list(one.materials), list(one.textures))
    from raytracer_amd.scene import CameraSettings

    tiny_cam = CameraSettings.look_at((0.0, 0.0, 3.0), (0.0, 0.0, 0.0), 50.0, z_near=1e-3)
    cases = [(cornell_with(cornell), None, RenderSettings(120, 90, 0, 50, 6, (0.2, 0.1, 0.3), F.RT_MODE_LINEAR_MEAN, 3)),
             (cornell, None, RenderSettings(31, 17, 0, 1, 3, mode=F.RT_MODE_LINEAR_MEAN, seed=5)),
             (cornell, None, RenderSettings(64, 48, 0, 2, 3, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=6)),
             (sph, sph_cam, RenderSettings(160, 90, 0, 50, 4, (0.7, 0.8, 1.0), F.RT_MODE_GAMMA_PER_SAMPLE, 6)),
             (lit, lit_cam, RenderSettings(160, 100, 0, 50, 3, mode=F.RT_MODE_LINEAR_MEAN, seed=7)),
             (soup, soup_cam, RenderSettings(128, 96, 0, 50, 3, mode=F.RT_MODE_LINEAR_MEAN, seed=4)),
             (mix_scene(cornell, lambda sc: (sc.mix(sc.lambertian((0.7, 0.6, 0.5)), sc.metal((0.9, 0.85, 0.8), 0.3), 0.5),
                                             sc.mix(sc.dielectric(1.5), sc.diffuse_light((1.0, 0.9, 0.8), 3.0), 0.35))), None,
              RenderSettings(96, 64, 0, 50, 4, (0.05, 0.1, 0.2), F.RT_MODE_LINEAR_MEAN, 22)),
             (one, tiny_cam, RenderSettings(48, 48, 0, 8, 4, (0.3, 0.4, 0.5), F.RT_MODE_LINEAR_MEAN, 2)),
             (two, tiny_cam, RenderSettings(48, 48, 0, 8, 4, (0.3, 0.4, 0.5), F.RT_MODE_GAMMA_PER_SAMPLE, 2))]
    for desc, cs, rs in cases:
        gpu, cpu, gst, cst = render_pair(ctx, desc, cs or desc.camera(0), rs)
        assert_images_equal(gpu, cpu)
        assert gst.rays_traced == cst.rays_traced and gst.pipeline == F.RT_PIPELINE_LANES
    ctx.close()


@pytest.mark.parametrize("knobs", [
    {"RT_LANES_TRAVERSAL": "walk", "RT_WALK_TOP_KB": "0"},
    {"RT_LANES_TRAVERSAL": "walk", "RT_WALK_TOP_KB": "1", "RT_WALK_THREADS": "256", "RT_WALK_BLOCKS_PER_CU": "4"},
    {"RT_LANES_TRAVERSAL": "walk", "RT_WALK_TOP_KB": "5", "RT_WALK_THREADS": "512", "RT_WALK_BLOCKS_PER_CU": "3"},
    {"RT_LANES_TRAVERSAL": "walk", "RT_WALK_TOP_KB": "150", "RT_WALK_THREADS": "1024", "RT_WALK_BLOCKS_PER_CU": "1"},
    {"RT_LANES_TRAVERSAL": "chain", "RT_CHAIN_TOP_RECORDS": "0"},
    {"RT_LANES_TRAVERSAL": "chain", "RT_CHAIN_TOP_RECORDS": "1", "RT_CHAIN_THREADS": "256"},
    {"RT_LANES_TRAVERSAL": "chain", "RT_CHAIN_TOP_RECORDS": "7", "RT_CHAIN_THREADS": "512"},
    {"RT_LANES_TRAVERSAL": "chain", "RT_CHAIN_TOP_RECORDS": "1200", "RT_CHAIN_THREADS": "256"},
], ids=lambda k: "-".join(v for v in k.values()))
def test_lds_prefix_sizes_and_block_shapes(cornell, teapot, monkeypatch, knobs):
    """The part of the tree kept in LDS is a performance choice and must never show in a result: the walk table's
    level-ordered prefix from none at all over one that a walk leaves and re-enters at almost every step (1 KB = 32
    nodes) to one that holds whole trees, and the chain records' LDS copies from none over the root alone to more than
    most trees have; each with a block shape other than the default.  Closest hits, sums and ray counts equal the
    oracle's bit for bit on Cornell, the teapot, a 60 000-triangle soup and three random scenes."""
    from helpers import random_scene

    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    ctx = Context(0)
    ctx.set_pipeline(F.RT_PIPELINE_LANES)
    lit, lit_cam = lit_teapot(teapot)
    soup, soup_cam = triangle_soup(60000, emissive_fraction=0.02)
    cases = [(cornell, None, (-1, -1, -1), (1, 1, 1)), (lit, lit_cam, (-35, -20, -22), (35, 18, 22)),
             (soup, soup_cam, (-1, -1, -1), (1, 1, 1))]
    cases += [random_scene(seed) + ((-2.5, -2.5, -2.5), (2.5, 2.5, 2.5)) for seed in (3, 31, 40)]
    for k, (desc, cs, lo, hi) in enumerate(cases):
        g, o = both(ctx, desc)
        rays = random_rays(20011, lo, hi, seed=900 + k, axis_aligned_fraction=0.2)
        got, _ = g.trace_closest(rays, 1e-3, np.inf)
        want, _ = o.trace_closest(rays, 1e-3, np.inf)
        assert_hits_equal(got, want)
        cam = camera_from_settings(cs or desc.camera(0), 96, 64)
        rs = RenderSettings(96, 64, 0, 10, 4, mode=F.RT_MODE_LINEAR_MEAN, seed=300 + k, background_color=(0.2, 0.3, 0.4))
        r = Renderer(cam, g, rs)
        gpu = r.render_sums()
        cpu, cst = o.render(cam, rs, threads=16)
        assert_images_equal(gpu, cpu)
        assert r.last_stats.rays_traced == cst.rays_traced and r.last_stats.pipeline == F.RT_PIPELINE_LANES
        g.close()
    ctx.close()


def test_mid_size_scene_uses_the_shared_lds_copy(gpu_ctx):
    """Scene tables whose LDS copy (48 bytes per node and per primitive) takes between 40 and 150 KB (here 1000
    triangles = 141 KB, and the 485 spheres = 68 KB) are staged in LDS once per 1024-thread block by the default
    pipeline; sums and ray counts equal the oracle's."""
    for desc, cs, bg in ((triangle_soup(1000, emissive_fraction=0.05) + ((0.3, 0.4, 0.5),)), (rtiow_final_scene() + ((0.7, 0.8, 1.0),))):
        rs = RenderSettings(200, 120, 0, 20, 5, mode=F.RT_MODE_LINEAR_MEAN, seed=8, background_color=bg)
        gpu, cpu, gst, cst = render_pair(gpu_ctx, desc, cs, rs)
        assert_images_equal(gpu, cpu)
        assert gst.rays_traced == cst.rays_traced and gst.pipeline == F.RT_PIPELINE_PATHS_LDS


def test_mix_materials(gpu_ctx, cornell):
    """Mix (src/material.rs:186-204): blended attenuation, one extra draw to pick the ray, children that do
    not scatter (a light), the factor 0 / 1 shortcuts and nesting four deep."""
    def build(sc):
        lam = sc.lambertian((0.7, 0.6, 0.5))
        met = sc.metal((0.9, 0.85, 0.8), 0.3)
        die = sc.dielectric(1.5)
        light = sc.diffuse_light((1.0, 0.9, 0.8), 3.0)
        nested = sc.mix(sc.mix(sc.mix(sc.mix(lam, met, 0.5), die, 0.25), light, 0.6), met, 0.7)
        room = sc.mix(sc.mix(lam, light, 0.5), sc.mix(die, met, 1.0), 0.35)
        return nested, room

    desc = mix_scene(cornell, build)
    for mode in (F.RT_MODE_LINEAR_MEAN, F.RT_MODE_GAMMA_PER_SAMPLE):
        rs = RenderSettings(140, 100, 0, 50, 6, mode=mode, seed=21, background_color=(0.05, 0.1, 0.2))
        gpu, cpu, gst, cst = render_pair(gpu_ctx, desc, desc.camera(0), rs)
        assert_images_equal(gpu, cpu)
        assert gst.rays_traced == cst.rays_traced
        assert gst.pipeline in (F.RT_PIPELINE_PATHS, F.RT_PIPELINE_PATHS_LDS)  # the one-launch pipeline carries Mix
    # Mix through every way of running a batch: the per-bounce fused kernels have no Mix instance and hand such
    # scenes to the split kernels; tables in HBM and in LDS
    rs = RenderSettings(96, 64, 0, 50, 4, mode=F.RT_MODE_LINEAR_MEAN, seed=22, background_color=(0.05, 0.1, 0.2))
    for pipeline, ran in ((F.RT_PIPELINE_PATHS, F.RT_PIPELINE_PATHS), (F.RT_PIPELINE_SPLIT, F.RT_PIPELINE_SPLIT),
                          (F.RT_PIPELINE_FUSED_LDS, F.RT_PIPELINE_SPLIT_LDS), (F.RT_PIPELINE_HYBRID, F.RT_PIPELINE_SPLIT)):
        ctx = Context(0)
        ctx.set_pipeline(pipeline)
        gpu, cpu, gst, cst = render_pair(ctx, desc, desc.camera(0), rs)
        assert_images_equal(gpu, cpu)
        assert gst.rays_traced == cst.rays_traced and gst.pipeline == ran
        ctx.close()
    # a Mix scene too large for one 256-thread block's LDS: the 1024-thread and the HBM instances
    big, cs = triangle_soup(1000, emissive_fraction=0.05)
    lam, met = big.lambertian((0.8, 0.7, 0.6)), big.metal((0.9, 0.9, 0.9), 0.2)
    mixed = big.mix(lam, met, 0.4)
    big.meshes[0] = TriangleMesh(big.meshes[0].positions, big.meshes[0].indices, big.meshes[0].normals, big.meshes[0].uvs, mixed)
    rs = RenderSettings(160, 100, 0, 20, 4, mode=F.RT_MODE_LINEAR_MEAN, seed=9, background_color=(0.3, 0.4, 0.5))
    for pipeline in (F.RT_PIPELINE_AUTO, F.RT_PIPELINE_PATHS):
        ctx = Context(0)
        ctx.set_pipeline(pipeline)
        gpu, cpu, gst, cst = render_pair(ctx, big, cs, rs)
        assert_images_equal(gpu, cpu)
        assert gst.rays_traced == cst.rays_traced
        assert gst.pipeline == (F.RT_PIPELINE_PATHS_LDS if pipeline == F.RT_PIPELINE_AUTO else F.RT_PIPELINE_PATHS)
        ctx.close()


def test_sphere_texture_coordinates(gpu_ctx):
    """Sphere::get_uv (src/object/sphere.rs:42-52): acos / atan2 on the GPU are restatements of glibc's acosf /
    atan2f, so image-textured spheres (the only consumers of sphere UVs) render bit for bit like the oracle,
    which calls the host libm.  Large textures make a one-ULP difference in u or v visible as a different texel;
    hit points off the unit sphere (radius != 1, centre != 0) feed acos arguments beyond [-1, 1] (NaN -> texel 0)."""
    rng = np.random.default_rng(5)
    sc = SceneDescription()
    for i, (c, r) in enumerate((((0.0, 0.0, 0.0), 1.0), ((2.2, 0.1, -0.5), 0.9), ((-2.1, -0.2, 0.3), 1.0), ((0.0, -101.0, 0.0), 100.0))):
        img = rng.integers(0, 256, size=(509 + i, 1021 - 3 * i, 3), dtype=np.uint8)
        sc.spheres.append(Sphere(c, r, sc.lambertian_texture(sc.add_texture(Texture.from_image(img)))))
    from raytracer_amd.scene import CameraSettings

    cs = CameraSettings.look_at((0.5, 1.5, 6.0), (0.0, 0.0, 0.0), 40.0, z_near=1e-3)
    rs = RenderSettings(320, 200, 0, 6, 6, mode=F.RT_MODE_LINEAR_MEAN, seed=12, background_color=(1.0, 1.0, 1.0))
    gpu, cpu, gst, cst = render_pair(gpu_ctx, sc, cs, rs)
    assert_images_equal(gpu, cpu)
    assert gst.rays_traced == cst.rays_traced
    assert len(np.unique(gpu.reshape(-1, 3), axis=0)) > 5000  # the texels show


def test_teapot_default_camera_is_black_and_lit_variant_matches(gpu_ctx, teapot):
    """Config 2 as the reference would run it: no camera, no light, background 0 -> identically zero image."""
    rs = RenderSettings(160, 90, 0, 50, 2, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=1)
    gpu, cpu, gst, cst = render_pair(gpu_ctx, teapot, teapot.camera(0), rs)
    assert_images_equal(gpu, cpu)
    assert not gpu.any()
    desc, cs = lit_teapot(teapot)
    rs = RenderSettings(200, 120, 0, 50, 4, mode=F.RT_MODE_LINEAR_MEAN, seed=1)
    gpu, cpu, gst, cst = render_pair(gpu_ctx, desc, cs, rs)
    assert_images_equal(gpu, cpu)
    assert gst.rays_traced == cst.rays_traced


def test_soup_render_small(gpu_ctx):
    """Config 5's generator at 100k triangles: geometric normals, emitters, incoherent bounces."""
    desc, cs = triangle_soup(100000, emissive_fraction=0.01)
    rs = RenderSettings(128, 96, 0, 50, 3, mode=F.RT_MODE_LINEAR_MEAN, seed=4)
    gpu, cpu, gst, cst = render_pair(gpu_ctx, desc, cs, rs)
    assert_images_equal(gpu, cpu)
    assert gst.rays_traced == cst.rays_traced


def test_image_texture(gpu_ctx, cornell):
    """Image texture on the Cornell walls (nearest texel, V flipped, column clamped with the height,
    src/texture.rs:115-136)."""
    desc = SceneDescription(list(cornell.meshes), [], list(cornell.materials), list(cornell.textures), list(cornell.cameras))
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(5, 9, 3), dtype=np.uint8)  # wider than tall: the clamp quirk bites
    m = desc.lambertian_texture(desc.add_texture(Texture.from_image(img)))
    base = desc.meshes[0]
    desc.meshes[0] = TriangleMesh(base.positions, base.indices, base.normals, base.uvs, m)
    rs = RenderSettings(128, 128, 0, 8, 4, mode=F.RT_MODE_LINEAR_MEAN, seed=6, background_color=(1, 1, 1))
    gpu, cpu, gst, cst = render_pair(gpu_ctx, desc, desc.camera(0), rs)
    assert_images_equal(gpu, cpu)
    empty = SceneDescription(list(desc.meshes), [], list(desc.materials), list(desc.textures), list(desc.cameras))
    empty.textures[-1] = Texture.from_image(np.zeros((0, 0, 3), np.uint8))  # empty image -> cyan
    gpu, cpu, _, _ = render_pair(gpu_ctx, empty, empty.camera(0), rs)
    assert_images_equal(gpu, cpu)


def test_png_textured_gltf_end_to_end(gpu_ctx, tmp_path):
    """A glTF file whose base-colour texture is a PNG, read by the library's own reader (rt_gltf_load + its PNG
    decoder) and rendered on the GPU, against the oracle fed by the Python reader: same sums bit for bit.  A
    light quad is added on both sides so that the texture colour reaches the image."""
    from helpers import encode_png
    from raytracer_amd.scene import CameraSettings, load_from_gltf, load_from_gltf_native
    from test_host import _textured_gltf

    rng = np.random.default_rng(21)
    png = encode_png(rng.integers(0, 256, (6, 11, 4)), 6, interlace=True)
    path = _textured_gltf(tmp_path, "textured", png, "glb")
    scenes = []
    for sc in (load_from_gltf_native(path), load_from_gltf(path)):
        light = sc.diffuse_light((1.0, 1.0, 1.0), 4.0)
        pos = np.array([[-2, -2, 1.5], [3, -2, 1.5], [3, 3, 1.5], [-2, 3, 1.5]], dtype=np.float32)
        sc.meshes.append(TriangleMesh(pos, np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32), np.zeros((0, 3), np.float32),
                                      np.zeros((0, 2), np.float32), light))
        scenes.append(sc)
    cs = CameraSettings.look_at((0.4, 0.4, 1.2), (0.3, 0.3, 0.0), 60.0, z_near=0.01)
    w, h = 96, 96
    rs = RenderSettings(w, h, 0, 6, 8, mode=F.RT_MODE_LINEAR_MEAN, seed=4, background_color=(0.1, 0.1, 0.1))
    cam = camera_from_settings(cs, w, h)
    gpu = Renderer(cam, Scene(gpu_ctx, scenes[0]), rs).render_sums()
    cpu, _ = ob.OracleScene(scenes[1]).render(cam, rs, threads=8)
    assert_images_equal(gpu, cpu)
    assert len(np.unique(gpu.reshape(-1, 3), axis=0)) > 50  # the texels show


@pytest.mark.parametrize("block", [0, 1, 2, 3])
def test_random_scenes_differential(gpu_ctx, block):
    """Differential test on random scenes (helpers.random_scene): every material and texture kind, Mix trees,
    spheres before or after the meshes, shared edges, duplicated and degenerate triangles, defocus.  Closest
    hits of incoherent rays and rendered sums must equal the oracle's bit for bit, with and without the BVH."""
    from helpers import random_scene

    for seed in range(block * 6, block * 6 + 6):
        desc, cs = random_scene(seed)
        g, o = both(gpu_ctx, desc)
        rays = random_rays(6000, (-2.5, -2.5, -2.5), (2.5, 2.5, 2.5), seed=seed, axis_aligned_fraction=0.2)
        for use_bvh in (True, False):
            got, _ = g.trace_closest(rays, 1e-3, np.inf, use_bvh=use_bvh)
            want, _ = o.trace_closest(rays, 1e-3, np.inf, use_bvh=use_bvh)
            assert_hits_equal(got, want)
        w, h = 72, 56
        cam = camera_from_settings(cs, w, h)
        mode = F.RT_MODE_GAMMA_PER_SAMPLE if seed % 2 else F.RT_MODE_LINEAR_MEAN
        rs = RenderSettings(w, h, 0, 12, 5, mode=mode, seed=100 + seed, background_color=(0.2, 0.3, 0.4))
        r = Renderer(cam, g, rs)
        gpu = r.render_sums()
        cpu, cst = o.render(cam, rs, threads=16)
        assert_images_equal(gpu, cpu)
        assert r.last_stats.rays_traced == cst.rays_traced


@pytest.mark.parametrize("traversal", ["walk", "chain"])
def test_random_scenes_with_the_lanes_traversals(monkeypatch, traversal):
    """The random scenes again (shared edges, duplicated and degenerate triangles, coplanar axis-aligned meshes whose
    boxes only differ by the 0.001 padding: exact ties, child boxes that do not share their parent's planes) through
    the one-lane-per-ray closest-hit kernel and the one-lane-per-path render pipeline, with either traversal forced.
    The chain records decide a node's grandchildren ahead of time and re-decide them from the stack; ties and
    padded boxes are where that could go wrong.  Hits, sums and ray counts equal the oracle's bit for bit."""
    from helpers import random_scene

    monkeypatch.setenv("RT_LANES_TRAVERSAL", traversal)
    ctx = Context(0)
    ctx.set_pipeline(F.RT_PIPELINE_LANES)
    for seed in range(24, 44):
        desc, cs = random_scene(seed)
        g, o = both(ctx, desc)
        rays = random_rays(5000, (-2.5, -2.5, -2.5), (2.5, 2.5, 2.5), seed=seed, axis_aligned_fraction=0.3)
        got, _ = g.trace_closest(rays, 1e-3, np.inf)
        want, _ = o.trace_closest(rays, 1e-3, np.inf)
        assert_hits_equal(got, want)
        cam = camera_from_settings(cs, 64, 48)
        rs = RenderSettings(64, 48, 0, 12, 4, mode=F.RT_MODE_LINEAR_MEAN, seed=200 + seed, background_color=(0.2, 0.3, 0.4))
        r = Renderer(cam, g, rs)
        gpu = r.render_sums()
        cpu, cst = o.render(cam, rs, threads=16)
        assert_images_equal(gpu, cpu)
        assert r.last_stats.rays_traced == cst.rays_traced and r.last_stats.pipeline == F.RT_PIPELINE_LANES
    ctx.close()


@pytest.mark.parametrize("name", ["cornell", "teapot", "spheres"])
def test_converged_images_match_the_reference_rng(gpu_ctx, cornell, teapot, name):
    """Converged-image parity at equal spp (256 x 256, 256 spp, progressive mode): the GPU image (counter stream)
    against the oracle rendering with the reference's generator, per-thread xoshiro256++ (src/random.rs:8-14,
    src/renderer.rs:105-165), under two seeds that also give the Monte-Carlo noise level.  Tolerances (stated in
    helpers.assert_image_within_tolerance): per-channel RMSE <= 3 sigma_MC over finite pixels, mean luminance
    within 1 % (or four standard errors where the image is noisier than that), inf-pixel fraction within 20 %."""
    from helpers import assert_image_within_tolerance

    if name == "cornell":
        desc, cs, bg = cornell, cornell.camera(0), (0.0, 0.0, 0.0)
    elif name == "teapot":
        desc, cs = lit_teapot(teapot)
        bg = (0.0, 0.0, 0.0)
    else:
        desc, cs = rtiow_final_scene()
        bg = (0.7, 0.8, 1.0)
    w = h = spp = 256
    g, o = both(gpu_ctx, desc)
    cam = camera_from_settings(cs, w, h)
    scale = np.float32(1.0) / np.float32(spp)
    gpu = Renderer(cam, g, RenderSettings(w, h, 0, 50, spp, bg, F.RT_MODE_GAMMA_PER_SAMPLE, 1)).render_sums() * scale
    cpu = [o.render(cam, RenderSettings(w, h, 0, 50, spp, bg, F.RT_MODE_GAMMA_PER_SAMPLE, seed), threads=16,
                    rng_mode=ob.RNG_THREAD)[0] * scale for seed in (11, 12)]
    rep = assert_image_within_tolerance(gpu, cpu[0], cpu[1])
    if name != "spheres":
        assert rep["inf"]["g"] > 0.01


def test_batching_and_sample_ranges_do_not_change_the_sums(gpu_ctx, cornell):
    """The image is a pure function of (seed, pixel, sample): splitting the work into small batches, or into
    consecutive sample ranges accumulated into one device buffer, gives bit-identical sums."""
    import torch

    w, h, spp = 96, 64, 10
    cam = camera_from_settings(cornell.camera(0), w, h)
    rs = RenderSettings(w, h, 0, 50, spp, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=11)
    scene = Scene(gpu_ctx, cornell)
    full = Renderer(cam, scene, rs).render_sums()
    gpu_ctx.set_max_paths(w * h * 3)  # 4 batches: 3 + 3 + 3 + 1 samples
    small = Renderer(cam, scene, rs).render_sums()
    gpu_ctx.set_max_paths(1000)       # fewer paths than pixels: one sample per batch
    tiny = Renderer(cam, scene, rs).render_sums()
    gpu_ctx.set_max_paths(0)
    assert bits_equal(full, small).all() and bits_equal(full, tiny).all()
    agg = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    lib = scene.lib
    for begin, count in ((0, 4), (4, 1), (5, 5)):
        s = RenderSettings(w, h, 0, 50, spp, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=11, sample_begin=begin,
                           sample_count=count).to_c()
        F.check(lib, lib.rt_render_device(gpu_ctx.handle, scene.handle, C.byref(cam), C.byref(s),
                                          C.c_void_p(agg.data_ptr()), None))
    gpu_ctx.synchronize()
    assert bits_equal(full.reshape(-1), agg.cpu().numpy()).all()


def test_render_progressive_reports(gpu_ctx, cornell):
    """Renderer.render_progressive reports after samples 1-5, every 16th and the last, each report being the
    running mean of gamma-mapped samples (src/renderer.rs:138-148)."""
    w, h, spp = 64, 48, 18
    cam = camera_from_settings(cornell.camera(0), w, h)
    scene, oracle = both(gpu_ctx, cornell)
    rs = RenderSettings(w, h, 0, 50, spp, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=3)
    reports = []
    Renderer(cam, scene, rs).render_progressive(lambda px, ww, hh: reports.append(px.copy()))
    assert len(reports) == 7  # samples 1,2,3,4,5,16,18
    for n, px in zip((1, 2, 3, 4, 5, 16, 18), reports):
        cpu, _ = oracle.render(cam, RenderSettings(w, h, 0, 50, spp, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=3,
                                                   sample_count=n), threads=8)
        assert bits_equal(px, cpu * (np.float32(1.0) / np.float32(n))).all()
    img8 = Renderer(cam, scene, rs).render()
    assert img8.dtype == np.uint8 and img8.shape == (h, w, 3)


def test_device_accumulator_helpers(gpu_ctx, cornell):
    """rt_accum_create / rt_accum_download / rt_accum_destroy: the running aggregate of render_progressive for hosts
    without a device allocator (the Rust patch, the C++ host).  Two calls over consecutive sample ranges into one
    accumulator give, bit for bit, the oracle's sums over the whole range, whatever the split."""
    lib = gpu_ctx.lib
    w, h, spp = 40, 24, 7
    cam = camera_from_settings(cornell.camera(0), w, h)
    scene, oracle = both(gpu_ctx, cornell)
    want, _ = oracle.render(cam, RenderSettings(w, h, 0, 50, spp, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=11), threads=8)
    for split in (1, 3, 6):
        d = C.c_void_p()
        F.check(lib, lib.rt_accum_create(gpu_ctx.handle, w, h, C.byref(d)))
        for begin, count in ((0, split), (split, spp - split)):
            rs = RenderSettings(w, h, 0, 50, spp, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=11, sample_begin=begin,
                                sample_count=count).to_c()
            F.check(lib, lib.rt_render_device(gpu_ctx.handle, scene.handle, C.byref(cam), C.byref(rs), d, None))
        got = np.zeros((h, w, 3), np.float32)
        F.check(lib, lib.rt_accum_download(gpu_ctx.handle, d, got.size, got.ctypes.data))
        F.check(lib, lib.rt_accum_destroy(gpu_ctx.handle, d))
        assert bits_equal(got, want).all(), split
    assert lib.rt_accum_create(gpu_ctx.handle, 0, 4, C.byref(C.c_void_p())) == F.RT_ERR_INVALID_ARGUMENT
    assert lib.rt_accum_destroy(gpu_ctx.handle, None) == F.RT_OK


def test_render_edge_cases(gpu_ctx, cornell):
    scene, oracle = both(gpu_ctx, cornell)
    for (w, h, depth, spp) in ((1, 1, 50, 3), (7, 3, 1, 2), (5, 5, 0, 2), (33, 17, 2, 1), (257, 1, 50, 2)):
        cam = camera_from_settings(cornell.camera(0), w, h)
        rs = RenderSettings(w, h, 0, depth, spp, mode=F.RT_MODE_LINEAR_MEAN, seed=8)
        r = Renderer(cam, scene, rs)
        gpu = r.render_sums()
        cpu, cst = oracle.render(cam, rs, threads=2)
        assert bits_equal(gpu, cpu).all(), (w, h, depth, spp)
        assert r.last_stats.rays_traced == cst.rays_traced


def test_degenerate_trees_through_the_lds_pipelines():
    """Scenes of one and two primitives (the root is a leaf / the only interior node) through the pipelines that walk an
    LDS copy of the tree (eight copies per node, 48-byte sorted nodes, the split kernels' 32-byte nodes) and AUTO."""
    from raytracer_amd.scene import CameraSettings

    one = SceneDescription()
    one.meshes.append(TriangleMesh(np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], np.float32), np.array([[0, 1, 2]], np.uint32),
                                   np.zeros((0, 3), np.float32), np.zeros((0, 2), np.float32), one.lambertian((0.5, 0.6, 0.7))))
    two = SceneDescription(list(one.meshes), [Sphere((0.0, 0.0, -1.0), 0.5, 0)], list(one.materials), list(one.textures))
    cs = CameraSettings.look_at((0.0, 0.0, 3.0), (0.0, 0.0, 0.0), 50.0, z_near=1e-3)
    rs = RenderSettings(48, 40, 0, 8, 3, mode=F.RT_MODE_LINEAR_MEAN, seed=31, background_color=(0.3, 0.2, 0.1))
    for pipeline in (F.RT_PIPELINE_AUTO, F.RT_PIPELINE_PATHS_LDS, F.RT_PIPELINE_FUSED_LDS, F.RT_PIPELINE_SPLIT_LDS):
        ctx = Context(0)
        ctx.set_pipeline(pipeline)
        for desc in (one, two):
            gpu, cpu, gst, cst = render_pair(ctx, desc, cs, rs)
            assert_images_equal(gpu, cpu)
            assert gst.rays_traced == cst.rays_traced
        ctx.close()


def test_error_codes_through_the_abi(gpu_ctx, cornell):
    lib = gpu_ctx.lib
    h = C.c_void_p()
    assert lib.rt_scene_create(gpu_ctx.handle, None, C.byref(h)) == F.RT_ERR_INVALID_ARGUMENT
    d, keep = SceneDescription().to_desc()
    assert lib.rt_scene_create(gpu_ctx.handle, C.byref(d), C.byref(h)) == F.RT_ERR_INVALID_ARGUMENT
    assert b"empty scene" in lib.rt_last_error()
    sc = SceneDescription()
    sc.materials.append(Material(F.RT_MAT_TROWBRIDGE_REITZ, -1, 0.3))
    sc.spheres.append(Sphere((0, 0, 0), 1.0, 0))
    with pytest.raises(F.RtError) as e:
        Scene(gpu_ctx, sc)
    assert e.value.code == F.RT_ERR_UNSUPPORTED and "todo!()" in e.value.message
    assert lib.rt_context_create(99, None, 0, C.byref(h)) == F.RT_ERR_NO_DEVICE
    scene = Scene(gpu_ctx, cornell)
    cam = camera_from_settings(cornell.camera(0), 8, 8)
    bad = RenderSettings(0, 8, 0, 50, 1).to_c()
    buf = np.zeros(8 * 8 * 3, np.float32)
    assert lib.rt_render(gpu_ctx.handle, scene.handle, C.byref(cam), C.byref(bad), buf.ctypes.data, None) == F.RT_ERR_INVALID_ARGUMENT
    bad = RenderSettings(8, 8, 0, 50, 1, mode=7).to_c()
    assert lib.rt_render(gpu_ctx.handle, scene.handle, C.byref(cam), C.byref(bad), buf.ctypes.data, None) == F.RT_ERR_INVALID_ARGUMENT
    assert lib.rt_render(gpu_ctx.handle, scene.handle, C.byref(cam), None, buf.ctypes.data, None) == F.RT_ERR_INVALID_ARGUMENT
    assert lib.rt_device_count() >= 1


def test_full_size_properties_cornell_1080p(gpu_ctx, cornell):
    """At the benchmark's size the oracle is too slow to compare every pixel; check size-independent
    properties instead: determinism, additivity over sample ranges, the ray-count identity, and agreement
    with the oracle on a strip of rows rendered at the same resolution."""
    w, h = 1920, 1080
    cam = camera_from_settings(cornell.camera(0), w, h)
    scene, oracle = both(gpu_ctx, cornell)
    rs = RenderSettings(w, h, 0, 50, 4, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=1)
    r = Renderer(cam, scene, rs)
    a = r.render_sums()
    st = r.last_stats
    b = r.render_sums()
    assert bits_equal(a, b).all(), "two runs must be bit-identical"
    assert st.paths == w * h * 4 and st.paths <= st.rays_traced <= 50 * st.paths
    assert 1.5 < st.rays_traced / st.paths < 3.0
    assert not np.isnan(a).any() and 0.001 < np.isinf(a).mean() < 0.1
    # the same pixels rendered by the oracle: full-resolution camera, rows 500..507 only
    sub = RenderSettings(w, h, 0, 50, 4, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=1)
    cpu, _ = oracle_rows(oracle, cam, sub, 500, 508)
    assert bits_equal(a[500:508], cpu).all()


def oracle_rows(oracle, cam, rs, row0, row1):
    """Oracle render of rows [row0, row1) of the full image: pixel indices (and hence random streams) are
    those of the full frame because the oracle is asked for the full frame's first row1 rows."""
    part = RenderSettings(rs.image_width, row1, 0, rs.max_depth, rs.samples_per_pixel, rs.background_color,
                          rs.mode, rs.seed, rs.sample_begin, rs.sample_count)
    img, st = oracle.render(cam, part, threads=16)
    return img[row0:row1], st


def test_config4_cornell_4k_split_over_eight_sample_blocks(gpu_ctx, cornell):
    """Cornell at 3840x2160 with the sample budget split into eight contiguous blocks (what eight GPUs
    render): adding the blocks one after the other into one device accumulator equals rendering all samples in
    one call bit for bit, in both modes, and a strip of rows equals the oracle's."""
    import torch

    from raytracer_amd.distributed import sample_range

    w, h, spp = 3840, 2160, 8
    cam = camera_from_settings(cornell.camera(0), w, h)
    scene, oracle = both(gpu_ctx, cornell)
    lib = scene.lib
    for mode in (F.RT_MODE_LINEAR_MEAN, F.RT_MODE_GAMMA_PER_SAMPLE):
        whole = Renderer(cam, scene, RenderSettings(w, h, 0, 50, spp, mode=mode, seed=2)).render_sums()
        buf = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
        for rank in range(8):
            begin, count = sample_range(spp, 8, rank)
            st = RenderSettings(w, h, 0, 50, spp, mode=mode, seed=2, sample_begin=begin, sample_count=count).to_c()
            F.check(lib, lib.rt_render_device(gpu_ctx.handle, scene.handle, C.byref(cam), C.byref(st), buf.data_ptr(), None))
        gpu_ctx.synchronize()
        assert bits_equal(buf.cpu().numpy().reshape(h, w, 3), whole).all()
    cpu, _ = oracle_rows(oracle, cam, RenderSettings(w, h, 0, 50, spp, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=2), 700, 702)
    assert bits_equal(whole[700:702], cpu).all()


def test_config2_teapot_and_config3_spheres_at_1080p(gpu_ctx, teapot):
    """The reference's teapot asset at 1920x1080 with its default camera has no emitter: the image is exactly
    zero and every path is traced to its end.  The 485-sphere scene at 1920x1080 equals the oracle on a strip
    of rows."""
    w, h = 1920, 1080
    scene = Scene(gpu_ctx, teapot)
    r = Renderer(camera_from_settings(teapot.camera(0), w, h), scene, RenderSettings(w, h, 0, 50, 2, seed=1))
    img = r.render_sums()
    assert not img.any() and r.last_stats.paths == w * h * 2 and r.last_stats.rays_traced > r.last_stats.paths
    desc, cs = rtiow_final_scene()
    g, o = both(gpu_ctx, desc)
    cam = camera_from_settings(cs, w, h)
    rs = RenderSettings(w, h, 0, 50, 2, mode=F.RT_MODE_LINEAR_MEAN, seed=3, background_color=(0.7, 0.8, 1.0))
    gpu = Renderer(cam, g, rs).render_sums()
    cpu, _ = oracle_rows(o, cam, rs, 600, 604)
    assert bits_equal(gpu[600:604], cpu).all()


def test_config5_soup_one_million_triangles_at_1080p(gpu_ctx):
    """The triangle soup of the memory-bound configuration at a tenth of its size (the oracle builds the
    10-million-triangle tree too slowly for a test): trees identical, 1080p render equal on a strip of rows."""
    desc, cs = triangle_soup(1_000_000)
    g, o = both(gpu_ctx, desc)
    gb, gp, gs = g.export_bvh()
    obx, op, os_ = o.export_bvh()
    assert np.array_equal(gb.view(np.uint32), obx.view(np.uint32)) and np.array_equal(gp, op) and np.array_equal(gs, os_)
    w, h = 1920, 1080
    cam = camera_from_settings(cs, w, h)
    rs = RenderSettings(w, h, 0, 50, 1, mode=F.RT_MODE_LINEAR_MEAN, seed=1)
    r = Renderer(cam, g, rs)
    gpu = r.render_sums()
    cpu, _ = oracle_rows(o, cam, rs, 540, 544)
    assert bits_equal(gpu[540:544], cpu).all()
    assert r.last_stats.rays_traced > 1.5 * r.last_stats.paths


def test_config5_soup_ten_million_triangles_full_size(gpu_ctx):
    """The memory-bound configuration at its full size: 10 000 000 triangles, 1920 x 1080.  The oracle cannot
    build this tree in test time, so the checks are size-independent properties: the device-built tree satisfies
    every structural rule of the reference's builder (src/bvh.rs:248-285, src/aabb.rs:78-93) and holds each
    primitive once; closest hits through the tree equal the tree-less linear scan (World::hit order) except where
    two triangles tie bit for bit; two renders are bit-identical; consecutive sample ranges add up to the whole;
    and every path is traced (ray-count identity)."""
    import torch

    from test_oracle import check_bvh_structure_vectorised

    n_tris = 10_000_000
    desc, cs = triangle_soup(n_tris)
    scene = Scene(gpu_ctx, desc)
    info = scene.info()
    assert info.num_triangles == n_tris and info.num_nodes == 2 * n_tris - 1 and 24 <= info.depth <= 25
    boxes, prim, skip = scene.export_bvh()
    check_bvh_structure_vectorised(boxes, prim, skip, n_tris)
    del boxes, prim, skip
    # BVH against the linear scan on incoherent rays: identical distances; identical primitives unless the scan
    # order and the tree order resolve an exact tie differently (none expected in a random soup)
    rays = random_rays(384, (-1, -1, -1), (1, 1, 1), seed=5, axis_aligned_fraction=0.1)
    tree, _ = scene.trace_closest(rays, 1e-4, np.inf)
    scan, _ = scene.trace_closest(rays, 1e-4, np.inf, use_bvh=False)
    assert np.array_equal(tree["t"].view(np.uint32), scan["t"].view(np.uint32))
    assert np.array_equal(tree["prim_id"], scan["prim_id"])
    assert np.array_equal(tree["normal"].view(np.uint32), scan["normal"].view(np.uint32))
    assert (tree["prim_id"] != F.RT_MISS).mean() > 0.8
    w, h = 1920, 1080
    cam = camera_from_settings(cs, w, h)
    rs = RenderSettings(w, h, 0, 50, 2, mode=F.RT_MODE_LINEAR_MEAN, seed=1)
    r = Renderer(cam, scene, rs)
    a = r.render_sums()
    st = r.last_stats
    assert st.pipeline == F.RT_PIPELINE_LANES
    b = r.render_sums()
    assert bits_equal(a, b).all(), "two runs must be bit-identical"
    assert st.paths == w * h * 2 and st.rays_traced == r.last_stats.rays_traced
    assert 1.5 * st.paths < st.rays_traced <= 50 * st.paths
    agg = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    rays_sum = 0
    for begin in (0, 1):
        part = RenderSettings(w, h, 0, 50, 2, mode=F.RT_MODE_LINEAR_MEAN, seed=1, sample_begin=begin, sample_count=1).to_c()
        pst = F.rt_stats()
        F.check(scene.lib, scene.lib.rt_render_device(gpu_ctx.handle, scene.handle, C.byref(cam), C.byref(part),
                                                      C.c_void_p(agg.data_ptr()), C.byref(pst)))
        rays_sum += pst.rays_traced
    gpu_ctx.synchronize()
    assert bits_equal(a.reshape(-1), agg.cpu().numpy()).all() and rays_sum == st.rays_traced
    scene.close()


def _read_pfm(path):
    with open(path, "rb") as fh:
        assert fh.readline().strip() == b"PF"
        w, h = (int(x) for x in fh.readline().split())
        assert float(fh.readline()) < 0  # little endian
        data = np.frombuffer(fh.read(), dtype="<f4").reshape(h, w, 3)
    return data[::-1]  # PFM stores rows bottom to top


def test_cpp_host_cli(cornell, tmp_path):
    """The C++ host layer (rt_host.hpp: load_from_gltf, Camera::new, Renderer::render / render_progressive,
    file ImageOutput) driven by rt_cli with the reference CLI's flags, against the oracle."""
    import subprocess

    cli = os.path.join(os.path.dirname(F.LIB_PATH), "rt_cli")
    asset = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "cornell.gltf")
    w, h, spp = 80, 60, 7
    oracle = ob.OracleScene(cornell)
    cam = ob.camera_from_settings(cornell.camera(0), w, h)
    out = str(tmp_path / "batch.pfm")
    r = subprocess.run([cli, "-W", str(w), "-H", str(h), "--spp", str(spp), "-d", "50", "-c", "0", "--seed", "9", "--batch",
                        asset, out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "building BVH took" in r.stdout and "Rendering took" in r.stdout
    cpu, _ = oracle.render(cam, RenderSettings(w, h, 0, 50, spp, mode=F.RT_MODE_LINEAR_MEAN, seed=9), threads=8)
    assert bits_equal(_read_pfm(out), cpu * (np.float32(1.0) / np.float32(spp))).all()
    out = str(tmp_path / "prog.pfm")
    r = subprocess.run([cli, "--width", str(w), "--height", str(h), "--spp", str(spp), "--seed", "9", "--debug",
                        "--bvh-disabled", asset, out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("Outputted image after sample") == 6  # samples 1,2,3,4,5,7
    cpu, _ = oracle.render(cam, RenderSettings(w, h, 0, 50, spp, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=9), threads=8)
    got, want = _read_pfm(out), cpu * (np.float32(1.0) / np.float32(spp))
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin) and np.allclose(got[fin], want[fin], rtol=1e-5, atol=1e-6)
    out = str(tmp_path / "img.ppm")
    r = subprocess.run([cli, "-W", "16", "-H", "12", "--spp", "2", asset, out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and open(out, "rb").read().startswith(b"P6\n16 12\n255\n")
    # PNG sink: the same 8-bit values as the PPM (pixels_to_image, src/renderer.rs:173-176)
    from raytracer_amd.png import decode_png_rgb8

    png_out, ppm_out = str(tmp_path / "img.png"), str(tmp_path / "same.ppm")
    for o in (png_out, ppm_out):
        r = subprocess.run([cli, "-W", "16", "-H", "12", "--spp", "2", asset, o], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
    ppm = np.frombuffer(open(ppm_out, "rb").read()[len(b"P6\n16 12\n255\n"):], dtype=np.uint8).reshape(12, 16, 3)
    assert np.array_equal(decode_png_rgb8(open(png_out, "rb").read()), ppm) and ppm.max() > 0
    r = subprocess.run([cli, "/nonexistent.gltf"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "error" in r.stderr


def test_single_process_node_entry_points(gpu_ctx, cornell, tmp_path):
    """rt_render_node / rt_allreduce_sum, the single-process multi-GPU entry points, on the one GPU this box
    has: a one-GPU node render must equal rt_render bit for bit, an all-reduce over one buffer (RCCL bound
    with dlopen) must leave it unchanged, and two contexts on one device are refused.  The N > 1 exchange is
    RCCL's own all-reduce and cannot run here."""
    import subprocess

    import torch

    from raytracer_amd.distributed import allreduce_sum, render_node

    scene = Scene(gpu_ctx, cornell)
    w, h, spp = 96, 64, 6
    cam = camera_from_settings(cornell.camera(0), w, h)
    for mode in (F.RT_MODE_LINEAR_MEAN, F.RT_MODE_GAMMA_PER_SAMPLE):
        rs = RenderSettings(w, h, 0, 50, spp, mode=mode, seed=5)
        want = Renderer(cam, scene, rs).render_sums()
        assert bits_equal(render_node([gpu_ctx], [scene], cam, rs), want).all()
        part = RenderSettings(w, h, 0, 50, spp, mode=mode, seed=5, sample_begin=2, sample_count=3)
        assert bits_equal(render_node([gpu_ctx], [scene], cam, part), Renderer(cam, scene, part).render_sums()).all()
    buf = torch.arange(1000, dtype=torch.float32, device="cuda") * 0.25
    keep = buf.clone()
    allreduce_sum([gpu_ctx], [buf.data_ptr()], buf.numel())
    gpu_ctx.synchronize()
    assert torch.equal(buf, keep)
    allreduce_sum([gpu_ctx], [0], 0)  # nothing to do
    other = Context(0)
    with pytest.raises(F.RtError) as e:
        allreduce_sum([gpu_ctx, other], [buf.data_ptr(), keep.data_ptr()], buf.numel())
    assert e.value.code == F.RT_ERR_INVALID_ARGUMENT and "share device" in e.value.message
    with pytest.raises(F.RtError):
        allreduce_sum([], [], 4)
    # the C++ host's --gpus path (NodeRenderer) against its single-context path
    cli = os.path.join(os.path.dirname(F.LIB_PATH), "rt_cli")
    asset = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "cornell.gltf")
    outs = []
    for extra in ([], ["--gpus", "1"]):
        out = str(tmp_path / f"node{len(outs)}.pfm")
        r = subprocess.run([cli, "-W", "64", "-H", "48", "--spp", "5", "--seed", "3", "--batch", *extra, asset, out],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        outs.append(_read_pfm(out))
    assert bits_equal(outs[0], outs[1]).all()


def test_bench_two_ranks_rehearsal(tmp_path):
    """bench.py's multi-rank path (sample sharding, double-buffered asynchronous all-reduce, max-over-ranks
    timing) rehearsed with two processes sharing the one GPU over gloo; the real thing runs one rank per GPU
    over RCCL.  The two-rank image must equal the one-rank image up to float association."""
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    size = ["--no-cpu-baseline", "--width", "320", "--height", "180", "--spp-per-step", "4", "--total-spp", "16"]
    common = ["--steps", "2", "--warmup", "1"] + size
    # two ranks x 2 steps x 4 spp and one rank x 4 steps x 4 spp both render samples [0, 16) of every pixel
    two_npz, one_npz = str(tmp_path / "two.npz"), str(tmp_path / "one.npz")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "2", "--backend", "gloo", "--share-device0", "--steps", "2", "--warmup", "0",
                          "--dump-image", two_npz] + size,
                         capture_output=True, text=True, timeout=280, cwd=root)
    assert two.returncode == 0, two.stderr[-2000:]
    # without a launcher bench.py starts the ranks itself; a launcher whose rank count contradicts --gpus is refused
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    own = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device0"] + common,
                         capture_output=True, text=True, timeout=280, cwd=root, env=env)
    assert own.returncode == 0, own.stderr[-2000:]
    jo = json.loads([l for l in own.stdout.splitlines() if l.startswith("{")][-1])
    assert jo["n_gpus"] == 2 and jo["ranks_in_process_group"] == 2
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"] + common, capture_output=True, text=True,
                         timeout=120, cwd=root, env={**env, "WORLD_SIZE": "2", "RANK": "0"})
    assert bad.returncode != 0 and "--gpus 4" in bad.stderr
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "0",
                          "--dump-image", one_npz] + size,
                         capture_output=True, text=True, timeout=280, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    j2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert j2["n_gpus"] == 2 and j2["ranks_in_process_group"] == 2 and j2["scaling"] == "weak" and j2["config"]["allreduce_floats_per_step"] == 320 * 180 * 3
    assert j1["n_gpus"] == 1 and j1["value"] > 0 and j2["value"] > 0
    assert j1["time_to_16spp_measured_s"] > 0 and j2["time_to_16spp_measured_s"] > 0
    # Same sample set, so the two-rank image is the one-rank image up to the association of the float additions:
    # through the timed steps (double-buffered asynchronous all-reduce per step) and through the measured
    # time-to-N-spp render (one exchange).  Infinite pixels (pdf = 0 bounces that met the light) and NaNs coincide.
    i1, i2 = np.load(one_npz), np.load(two_npz)
    for key in ("steps_mean", "total_spp_mean"):
        a, b = i1[key].reshape(-1), i2[key].reshape(-1)
        assert a.size == 320 * 180 * 3 and a.shape == b.shape
        assert np.array_equal(np.isinf(a), np.isinf(b)) and np.array_equal(np.isnan(a), np.isnan(b)), key
        fin = np.isfinite(a)
        assert fin.mean() > 0.9
        # sums of non-negative terms: no cancellation, so a few units in the last place; tolerance 2e-6 relative
        rel = np.abs(a[fin] - b[fin]) / np.maximum(np.abs(a[fin]), 1e-3)
        assert rel.max() <= 2e-6, (key, rel.max())
    assert np.abs(i1["steps_mean"].reshape(-1)[np.isfinite(i1["steps_mean"].reshape(-1))]).max() > 0.1
    assert abs(j1["rays_per_path"] - j2["rays_per_path"]) < 1e-9
    assert j1["roofline"]["algorithmic_gbs"] > 0 and j1["roofline"]["bound"] == "valu" and "cpu_baseline" not in j2
    assert j1["roofline"]["pmc"] is None and j1["roofline"]["traffic"] is None  # no profile speaks for a rehearsal size
