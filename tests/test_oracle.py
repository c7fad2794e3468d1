"""CPU tests of the oracle (the checker): known-answer rays, BVH structure, reference quirks, RNG."""
import json
import os

import numpy as np
import pytest

from helpers import assert_hits_equal, bits_equal, brute_force_f64, primary_rays, random_rays
from oracle import binding as ob
from raytracer_amd import RenderSettings, SceneDescription, Sphere, _ffi as F
from raytracer_amd.scene import Material, Texture, TriangleMesh
from raytracer_amd.scenes import rtiow_final_scene, triangle_soup

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_kat():
    with open(os.path.join(ROOT, "tests", "golden", "cornell_kat.json")) as fh:
        kat = json.load(fh)
    rays = np.zeros(len(kat["rays"]), dtype=F.RAY_DTYPE)
    for i, r in enumerate(kat["rays"]):
        rays["origin"][i] = r["origin"]
        rays["direction"][i] = r["direction"]
    return kat, rays


def check_against_kat(kat, hits):
    """The north star's bar: primitive ID exact (any member of the tie set on a shared edge is the closest
    primitive; the stored oracle answer pins which one), t within 1e-5 relative."""
    for r, h in zip(kat["rays"], hits):
        if not r["expected_prims"]:
            assert h["prim_id"] == F.RT_MISS, r["label"]
            continue
        assert int(h["prim_id"]) in r["expected_prims"], r["label"]
        assert abs(float(h["t"]) - r["expected_t_f64"]) <= 1e-5 * abs(r["expected_t_f64"]), r["label"]
        assert int(h["prim_id"]) == r["oracle_prim"], r["label"]
        assert int(np.float32(h["t"]).view(np.uint32)) == r["oracle_t_bits"], r["label"]


def test_known_answer_rays(cornell):
    kat, rays = load_kat()
    hits, _ = ob.OracleScene(cornell).trace_closest(rays, kat["t_min"], kat["t_max"])
    check_against_kat(kat, hits)


def test_oracle_matches_independent_float64_bruteforce(cornell):
    """Random rays inside the box: the oracle's primitive must be one of the float64 closest candidates and t
    must agree to 1e-5 relative, for both the BVH walk and the flat World scan."""
    o = ob.OracleScene(cornell)
    rays = random_rays(4000, (-0.95, -0.95, -0.95), (0.95, 0.95, 3.0), seed=11)
    t64, cands = brute_force_f64(cornell, rays, 1e-3, 1e9)
    for use_bvh in (True, False):
        hits, _ = o.trace_closest(rays, 1e-3, 1e9, use_bvh=use_bvh)
        bad = 0
        for i in range(len(rays)):
            if not cands[i]:
                bad += hits["prim_id"][i] != F.RT_MISS
            else:
                # grazing hits can fall either side of an edge in binary32; demand agreement in t, and the
                # primitive only where float64 is unambiguous about the neighbourhood
                ok_t = abs(float(hits["t"][i]) - t64[i]) <= 1e-5 * abs(t64[i])
                bad += not (ok_t and int(hits["prim_id"][i]) in cands[i])
        assert bad <= 2, f"{bad} disagreements with the float64 brute force (use_bvh={use_bvh})"


def test_bvh_and_world_agree_on_distance(cornell, teapot):
    """BvhNode::hit and World::hit find the same closest distance; the primitive may differ only on ties."""
    for desc, lo, hi in ((cornell, (-1, -1, -1), (1, 1, 3)), (teapot, (-30, -20, -20), (30, 20, 20))):
        o = ob.OracleScene(desc)
        rays = random_rays(1500, lo, hi, seed=5)
        a, _ = o.trace_closest(rays, 1e-4, np.inf, use_bvh=True)
        b, _ = o.trace_closest(rays, 1e-4, np.inf, use_bvh=False)
        same_t = bits_equal(a["t"], b["t"])
        assert same_t.mean() > 0.999
        differ = a["prim_id"] != b["prim_id"]
        assert np.all(bits_equal(a["t"], b["t"])[differ] | ~same_t[differ])


def test_tie_break_follows_visit_order(cornell):
    """On the back wall's shared diagonal both triangles are hit at the same t (edges are inclusive,
    src/object/triangle_mesh.rs:157-168).  World::hit keeps the later object (1); the BVH keeps whichever it
    visits last, which with this build rule is 0."""
    kat, rays = load_kat()
    o = ob.OracleScene(cornell)
    bvh, _ = o.trace_closest(rays[:1], kat["t_min"], kat["t_max"], use_bvh=True)
    world, _ = o.trace_closest(rays[:1], kat["t_min"], kat["t_max"], use_bvh=False)
    assert world["prim_id"][0] == 1 and bvh["prim_id"][0] == 0
    assert bits_equal(bvh["t"], world["t"]).all()


def check_bvh_structure(boxes, prim, skip, n_prims):
    n = len(prim)
    assert n == 2 * n_prims - 1
    assert skip[0] == n
    leaves = prim[prim >= 0]
    assert sorted(leaves.tolist()) == list(range(n_prims)), "every primitive in exactly one leaf"
    for i in range(n):
        if prim[i] >= 0:
            assert skip[i] == i + 1
        else:
            left, right = i + 1, skip[i + 1]
            assert i + 1 < right < skip[i] <= n
            assert skip[right] == skip[i]
            for c in (left, right):  # children inside the parent box
                assert np.all(boxes[c, :3] >= boxes[i, :3]) and np.all(boxes[c, 3:] <= boxes[i, 3:])
            sub = (skip[i] - i)
            assert sub % 2 == 1
            n_leaves = (sub + 1) // 2
            left_leaves = (right - left + 1) // 2
            assert left_leaves == n_leaves // 2, "mid = len / 2 (src/bvh.rs:268)"
    assert np.all(boxes[:, 3:] - boxes[:, :3] >= 0.00099), "pad_to_minimums (src/aabb.rs:78-93)"


def check_bvh_structure_vectorised(boxes, prim, skip, n_prims):
    """check_bvh_structure for trees too large for a Python loop (same assertions, numpy over all nodes)."""
    n = len(prim)
    assert n == 2 * n_prims - 1 and skip[0] == n
    leaf = prim >= 0
    assert np.array_equal(np.sort(prim[leaf]), np.arange(n_prims)), "every primitive in exactly one leaf"
    idx = np.arange(n, dtype=np.int64)
    sk = skip.astype(np.int64)
    assert np.array_equal(sk[leaf], idx[leaf] + 1)
    i = idx[~leaf]
    left = i + 1
    right = sk[left]
    assert np.all((left < right) & (right < sk[i]) & (sk[i] <= n))
    assert np.array_equal(sk[right], sk[i])
    for c in (left, right):
        assert np.all(boxes[c, :3] >= boxes[i, :3]) and np.all(boxes[c, 3:] <= boxes[i, 3:])
    sub = sk[i] - i
    assert np.all(sub % 2 == 1)
    assert np.array_equal((right - left + 1) // 2, ((sub + 1) // 2) // 2), "mid = len / 2 (src/bvh.rs:268)"
    assert np.all(boxes[:, 3:] - boxes[:, :3] >= 0.00099), "pad_to_minimums (src/aabb.rs:78-93)"


def test_bvh_structure(cornell, teapot):
    for desc in (cornell, teapot):
        o = ob.OracleScene(desc)
        boxes, prim, skip = o.export_bvh()
        check_bvh_structure(boxes, prim, skip, desc.num_primitives)
        check_bvh_structure_vectorised(boxes, prim, skip, desc.num_primitives)
    i = ob.OracleScene(cornell).info()
    assert (i.num_triangles, i.num_nodes, i.depth) == (36, 71, 7)
    # root box = padded union of the scene (src/main.rs:92-95)
    assert np.allclose(list(i.bounds_min), [-1.001, -1.00378394, -1.001], atol=1e-6)


def test_descending_sort_left_half_holds_larger_mins():
    """src/bvh.rs:264-270: the sort is DESCENDING by box minimum, so the left child gets the larger
    coordinates; equal keys are ordered by ascending primitive ID (our documented tie rule)."""
    sc = SceneDescription()
    m = sc.lambertian((0.5, 0.5, 0.5))
    for x in (0.0, 10.0, 20.0, 30.0):
        sc.spheres.append(Sphere((x, 0.0, 0.0), 1.0, m))
    boxes, prim, skip = ob.OracleScene(sc).export_bvh()
    assert prim[prim >= 0].tolist() == [3, 2, 1, 0]
    sc2 = SceneDescription()
    m = sc2.lambertian((0.5, 0.5, 0.5))
    for y in (0.0, 5.0, 7.0, 9.0):
        sc2.spheres.append(Sphere((0.0, y, 50.0), 1.0, m))  # identical x and z extents
    # longest_axis never compares x with y (src/aabb.rs:129-137): sizes x=2, y=11, z=2 -> x>z false, y>z true
    boxes, prim, skip = ob.OracleScene(sc2).export_bvh()
    assert prim[prim >= 0].tolist() == [3, 2, 1, 0]
    sc3 = SceneDescription()
    m = sc3.lambertian((0.5, 0.5, 0.5))
    for k in range(4):
        sc3.spheres.append(Sphere((0.0, 0.0, 0.0), 1.0 + k, m))  # distinct mins: -1, -2, -3, -4 -> order 0,1,2,3
    for k in range(3):
        sc3.spheres.append(Sphere((0.0, 0.0, 0.0), 1.0, m))      # three exact ties with sphere 0
    boxes, prim, skip = ob.OracleScene(sc3).export_bvh()
    # keys (box min): -1 for spheres 0, 4, 5, 6, then -2, -3, -4; ties in ascending ID at every level
    assert prim[prim >= 0].tolist() == [0, 4, 5, 6, 1, 2, 3]


def test_interval_semantics_per_primitive():
    """Triangles accept t == t_min / t == t_max (Range::contains), spheres do not (Range::surrounds)
    (src/object/triangle_mesh.rs:168, src/object/sphere.rs:67-70)."""
    sc = SceneDescription()
    m = sc.lambertian((0.5, 0.5, 0.5))
    pos = np.array([[-1, -1, -2], [1, -1, -2], [0, 1, -2]], dtype=np.float32)
    sc.meshes.append(TriangleMesh(pos, np.array([[0, 1, 2]], np.uint32), np.zeros((0, 3), np.float32),
                                  np.zeros((0, 2), np.float32), m))
    sc.spheres.append(Sphere((10.0, 0.0, -3.0), 1.0, m))
    o = ob.OracleScene(sc)
    rays = np.zeros(2, dtype=F.RAY_DTYPE)
    rays["direction"][0] = (0, 0, -1)         # triangle at t = 2 exactly
    rays["origin"][1] = (10, 0, 0)
    rays["direction"][1] = (0, 0, -1)         # sphere roots at t = 2 and t = 4 exactly
    h, _ = o.trace_closest(rays, 2.0, 100.0)
    assert h["prim_id"][0] == 0 and h["t"][0] == 2.0          # inclusive lower end
    assert h["prim_id"][1] == 1 and h["t"][1] == 4.0          # t == t_min rejected, far root taken
    h, _ = o.trace_closest(rays, 0.5, 2.0)
    assert h["prim_id"][0] == 0                                # inclusive upper end
    assert h["prim_id"][1] == F.RT_MISS                        # exclusive upper end
    # geometric normal (no vertex normals), face-forwarded against the ray
    h, _ = o.trace_closest(rays[:1], 0.1, 100.0)
    assert np.allclose(h["normal"][0], [0, 0, 1]) and h["front_facing"][0] == 1


def test_degenerate_sampler_fraction_and_inf_pixels(cornell):
    """src/sample.rs:13-22 swaps radius and angle: about 43 % of diffuse samples get z = 0 and pdf = 0, which
    turns beta into inf/NaN (src/renderer.rs:51-53).  The render must therefore contain +inf pixels and, in
    GAMMA mode, no NaN (linear_to_gamma maps NaN to 0, src/renderer.rs:159-165)."""
    o = ob.OracleScene(cornell)
    cam = ob.camera_from_settings(cornell.camera(0), 96, 96)
    rs = RenderSettings(96, 96, 0, 50, 8, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=3)
    img, st = o.render(cam, rs, threads=8)
    assert not np.isnan(img).any()
    assert 0.001 < np.isinf(img).mean() < 0.2
    assert np.all(img[np.isfinite(img)] >= 0)
    rs_lin = RenderSettings(96, 96, 0, 50, 8, mode=F.RT_MODE_LINEAR_MEAN, seed=3)
    lin, _ = o.render(cam, rs_lin, threads=8)
    assert np.isnan(lin).any(), "LINEAR mode keeps the NaNs the reference produces"
    assert 1.5 < st.rays_traced / st.paths < 3.5


def test_background_overwrites_radiance(cornell):
    """src/renderer.rs:59-63: a miss ASSIGNS the background; with a constant background every escaping path
    returns exactly that constant, so no pixel component can lie strictly between 0 and the constant unless
    it came from an emitter."""
    o = ob.OracleScene(cornell)
    cam = ob.camera_from_settings(cornell.camera(0), 48, 48)
    rs = RenderSettings(48, 48, 0, 50, 1, background_color=(0.25, 0.5, 0.75), mode=F.RT_MODE_LINEAR_MEAN, seed=9)
    img, _ = o.render(cam, rs, threads=4)
    px = img.reshape(-1, 3)
    is_bg = np.all(px == np.array([0.25, 0.5, 0.75], dtype=np.float32), axis=1)
    assert is_bg.mean() > 0.3


def test_render_is_deterministic_and_thread_count_independent(cornell):
    o = ob.OracleScene(cornell)
    cam = ob.camera_from_settings(cornell.camera(0), 40, 30)
    rs = RenderSettings(40, 30, 0, 50, 4, seed=42)
    a, _ = o.render(cam, rs, threads=1)
    b, _ = o.render(cam, rs, threads=7)
    assert bits_equal(a, b).all()
    c, _ = o.render(cam, RenderSettings(40, 30, 0, 50, 4, seed=43), threads=4)
    assert not bits_equal(a, c).all()


def test_sample_ranges_add_up(cornell):
    """Samples [0, 8) == samples [0, 3) then [3, 8) accumulated into the same buffer, bit for bit."""
    o = ob.OracleScene(cornell)
    cam = ob.camera_from_settings(cornell.camera(0), 32, 32)
    full, _ = o.render(cam, RenderSettings(32, 32, 0, 50, 8, seed=1), threads=4)
    part, _ = o.render(cam, RenderSettings(32, 32, 0, 50, 8, seed=1, sample_begin=0, sample_count=3), threads=4)
    part, _ = o.render(cam, RenderSettings(32, 32, 0, 50, 8, seed=1, sample_begin=3, sample_count=5), threads=4,
                       accum=part)
    assert bits_equal(full, part).all()


def test_thread_rng_mode_is_statistically_consistent(cornell):
    """The counter stream against the xoshiro256++ per-thread stream (the reference's SmallRng algorithm,
    src/random.rs:8-14) at equal spp: per-channel RMSE, mean luminance and inf-pixel fraction within the stated
    tolerance (helpers.assert_image_within_tolerance), the thread-stream noise level taken from two seeds.  The
    GPU renders with the counter stream bit for bit like the oracle in this mode, so this is the CPU half of the
    converged-image comparison; the GPU half is test_gpu_parity.test_converged_images_match_the_reference_rng."""
    from helpers import assert_image_within_tolerance

    o = ob.OracleScene(cornell)
    w = h = 96
    spp = 96
    cam = ob.camera_from_settings(cornell.camera(0), w, h)
    imgs = []
    for mode, seed in ((ob.RNG_COUNTER, 5), (ob.RNG_THREAD, 6), (ob.RNG_THREAD, 7)):
        rs = RenderSettings(w, h, 0, 50, spp, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=seed)
        img, _ = o.render(cam, rs, threads=8, rng_mode=mode)
        imgs.append(img / np.float32(spp))
    rep = assert_image_within_tolerance(*imgs)
    assert rep["inf"]["g"] > 0.01  # the reference's degenerate samples do poison pixels (SmallRng or not)


def test_rng_stream_properties():
    d = ob.rng_draws(1, 123, 7, 4096)
    assert np.array_equal(d, ob.rng_draws(1, 123, 7, 4096))
    assert not np.array_equal(d[:64], ob.rng_draws(1, 124, 7, 64))
    assert not np.array_equal(d[:64], ob.rng_draws(1, 123, 8, 64))
    assert not np.array_equal(d[:64], ob.rng_draws(2, 123, 7, 64))
    u = (d >> 8).astype(np.float32) * np.float32(2.0 ** -24)
    assert u.min() >= 0.0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.02
    bits = np.unpackbits(d.view(np.uint8))
    assert abs(bits.mean() - 0.5) < 0.01


def test_max_depth_edges(cornell):
    o = ob.OracleScene(cornell)
    cam = ob.camera_from_settings(cornell.camera(0), 16, 16)
    z, st = o.render(cam, RenderSettings(16, 16, 0, 0, 2, seed=1), threads=2)
    assert st.rays_traced == 0 and not z.any()
    one, st = o.render(cam, RenderSettings(16, 16, 0, 1, 2, mode=F.RT_MODE_LINEAR_MEAN, seed=1), threads=2)
    assert st.rays_traced == st.paths  # exactly one segment per path
    # with one segment only directly visible emitters contribute: values are 0 or 200 per sample
    vals = np.unique(one)
    assert set(vals.tolist()) <= {0.0, 200.0, 400.0}


def test_scene_validation_errors(cornell):
    with pytest.raises(F.RtError) as e:
        ob.OracleScene(SceneDescription())
    assert e.value.code == F.RT_ERR_INVALID_ARGUMENT
    sc = SceneDescription()
    sc.materials.append(Material(F.RT_MAT_TROWBRIDGE_REITZ, -1, 0.3))
    sc.spheres.append(Sphere((0, 0, 0), 1.0, 0))
    with pytest.raises(F.RtError) as e:
        ob.OracleScene(sc)
    assert e.value.code == F.RT_ERR_UNSUPPORTED
    sc = SceneDescription()
    sc.spheres.append(Sphere((0, 0, 0), 1.0, 3))
    with pytest.raises(F.RtError):
        ob.OracleScene(sc)
    sc = SceneDescription()
    a = sc.add_texture(Texture.checkerboard(1.0, 0, 0))  # refers to itself
    sc.materials.append(Material(F.RT_MAT_LAMBERTIAN, a))
    sc.spheres.append(Sphere((0, 0, 0), 1.0, 0))
    with pytest.raises(F.RtError):
        ob.OracleScene(sc)


def test_procedural_scenes_are_reproducible():
    a, cam_a = rtiow_final_scene()
    b, _ = rtiow_final_scene()
    assert 450 < len(a.spheres) < 500
    assert [s.center for s in a.spheres] == [s.center for s in b.spheres]
    kinds = [a.materials[s.material].kind for s in a.spheres]
    assert kinds.count(F.RT_MAT_METAL) > 20 and kinds.count(F.RT_MAT_DIELECTRIC) > 5
    s1, _ = triangle_soup(5000)
    s2, _ = triangle_soup(5000)
    assert np.array_equal(s1.meshes[0].positions, s2.meshes[0].positions)
    assert s1.num_primitives == 5000 and len(s1.meshes) == 2
    o = ob.OracleScene(s1)
    assert o.info().num_nodes == 9999


def mix_scene(cornell, build):
    """Cornell with the big box (mesh 4) and the back/floor/ceiling mesh (mesh 0) given materials from `build`."""
    sc = SceneDescription(list(cornell.meshes), [], list(cornell.materials), list(cornell.textures), list(cornell.cameras))
    m_box, m_room = build(sc)
    for idx, m in ((4, m_box), (0, m_room)):
        b = sc.meshes[idx]
        sc.meshes[idx] = TriangleMesh(b.positions, b.indices, b.normals, b.uvs, m)
    return sc


def test_mix_material_semantics(cornell):
    """Mix::scatter (src/material.rs:186-204): factor 1 -> right alone, factor 0 -> left alone (bit-identical
    images to using that material directly); other factors scatter both children and draw once more."""
    cam = ob.camera_from_settings(cornell.camera(0), 48, 48)
    rs = RenderSettings(48, 48, 0, 50, 4, mode=F.RT_MODE_LINEAR_MEAN, seed=12, background_color=(0.3, 0.3, 0.3))

    def render(build):
        img, st = ob.OracleScene(mix_scene(cornell, build)).render(cam, rs, threads=4)
        return img, st

    plain, _ = render(lambda sc: (sc.metal((0.9, 0.8, 0.7), 0.2), sc.lambertian((0.5, 0.6, 0.7))))
    right, _ = render(lambda sc: (sc.mix(sc.dielectric(1.5), sc.metal((0.9, 0.8, 0.7), 0.2), 1.0),
                                  sc.mix(sc.lambertian((0.5, 0.6, 0.7)), sc.dielectric(1.3), 0.0)))
    assert bits_equal(plain, right).all()
    both, st = render(lambda sc: (sc.mix(sc.dielectric(1.5), sc.metal((0.9, 0.8, 0.7), 0.2), 0.4),
                                  sc.lambertian((0.5, 0.6, 0.7))))
    assert not bits_equal(plain, both).all() and st.rays_traced > st.paths
    with pytest.raises(F.RtError):  # nesting deeper than 4
        def deep(sc):
            m = sc.lambertian((0.5, 0.5, 0.5))
            for _ in range(5):
                m = sc.mix(m, sc.dielectric(1.5), 0.5)
            return m, m
        render(deep)


def test_random_scenes_bvh_against_world_scan():
    """On random scenes (helpers.random_scene) the BVH walk and the linear scan of `World::hit` find the same
    distance for every ray, and the same primitive unless two primitives are hit at the same distance."""
    from helpers import random_rays, random_scene

    for seed in range(8):
        desc, _ = random_scene(seed)
        o = ob.OracleScene(desc)
        rays = random_rays(3000, (-2.5, -2.5, -2.5), (2.5, 2.5, 2.5), seed=seed, axis_aligned_fraction=0.2)
        a, _ = o.trace_closest(rays, 1e-3, np.inf, use_bvh=True)
        b, _ = o.trace_closest(rays, 1e-3, np.inf, use_bvh=False)
        assert np.array_equal(a["t"].view(np.uint32), b["t"].view(np.uint32))
        differ = a["prim_id"] != b["prim_id"]
        assert differ.mean() < 0.02  # exact ties only (shared edges, duplicated triangles)
        assert (a["prim_id"] != F.RT_MISS).mean() > 0.02
