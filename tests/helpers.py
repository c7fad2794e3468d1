"""Shared test utilities: ray generators and comparison helpers."""
import numpy as np

from raytracer_amd import _ffi as F


def primary_rays(cam, width, height, jitter_seed=None):
    """Camera rays through pixel centres (or jittered inside the pixel), un-normalised like Camera::get_ray."""
    jj, ii = np.meshgrid(np.arange(height, dtype=np.float32), np.arange(width, dtype=np.float32), indexing="ij")
    ii = ii.reshape(-1, 1)
    jj = jj.reshape(-1, 1)
    if jitter_seed is not None:
        rng = np.random.default_rng(jitter_seed)
        ii = ii + rng.random(ii.shape, dtype=np.float32) - np.float32(0.5)
        jj = jj + rng.random(jj.shape, dtype=np.float32) - np.float32(0.5)
    p00, du, dv, c = (np.array(list(v), dtype=np.float32) for v in
                      (cam.pixel_00_loc, cam.pixel_delta_u, cam.pixel_delta_v, cam.center))
    rays = np.zeros(width * height, dtype=F.RAY_DTYPE)
    rays["origin"] = c
    rays["direction"] = (p00 + du * ii + dv * jj) - c
    return rays


def random_rays(n, lo, hi, seed, axis_aligned_fraction=0.1):
    """Rays with origins uniform in the box [lo, hi] and random directions; a fraction gets exact zeros in
    one or two direction components (the reference's diffuse sampler produces d.z == 0 for ~43 % of bounces)."""
    rng = np.random.default_rng(seed)
    lo = np.asarray(lo, dtype=np.float32)
    hi = np.asarray(hi, dtype=np.float32)
    rays = np.zeros(n, dtype=F.RAY_DTYPE)
    rays["origin"] = (lo + (hi - lo) * rng.random((n, 3), dtype=np.float32)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    k = int(n * axis_aligned_fraction)
    if k:
        which = rng.integers(0, 3, size=k)
        d[np.arange(k), which] = 0.0
        d[: k // 4, (which[: k // 4] + 1) % 3] = 0.0
    rays["direction"] = d
    return rays


def bits_equal(a, b):
    """Element-wise: same bit pattern, or both NaN."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def assert_hits_equal(got, want):
    """Primitive ID bit-exact; t, u, v and normal bit-exact as well (the bar the north star sets is 1e-5
    relative for t and normals — bit equality is stricter and is what the implementation achieves)."""
    assert np.array_equal(got["prim_id"], want["prim_id"]), (
        f"{np.count_nonzero(got['prim_id'] != want['prim_id'])} primitive IDs differ")
    for f in ("t", "u", "v", "normal"):
        ok = bits_equal(got[f], want[f])
        assert ok.all(), f"field {f}: {np.count_nonzero(~ok)} values differ"
    assert np.array_equal(got["front_facing"], want["front_facing"])


def scene_triangles(desc):
    """(n, 3, 3) float64 vertices of every triangle in primitive-ID order (meshes in order, spheres excluded)."""
    tris = []
    for m in desc.meshes:
        tris.append(m.positions.astype(np.float64)[m.indices.astype(np.int64)])
    return np.concatenate(tris, axis=0) if tris else np.zeros((0, 3, 3))


def brute_force_f64(desc, rays, t_min, t_max, rel_tie=1e-6):
    """Independent float64 closest-hit over all triangles (Möller–Trumbore, no culling, edges inclusive).

    Returns (t, candidates): t[i] is the closest distance (inf on a miss) and candidates[i] the set of
    primitive IDs whose distance is within rel_tie of it (more than one on shared edges)."""
    tri = scene_triangles(desc)
    o = rays["origin"].astype(np.float64)
    d = rays["direction"].astype(np.float64)
    v0 = tri[:, 0][None]
    e1 = (tri[:, 1] - tri[:, 0])[None]
    e2 = (tri[:, 2] - tri[:, 0])[None]
    p = np.cross(d[:, None, :], e2)
    det = (e1 * p).sum(-1)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / det
        tv = o[:, None, :] - v0
        u = (tv * p).sum(-1) * inv
        q = np.cross(tv, e1)
        v = (d[:, None, :] * q).sum(-1) * inv
        t = (e2 * q).sum(-1) * inv
    eps = 1e-9
    ok = (np.abs(det) > 1e-12) & (u >= -eps) & (u <= 1 + eps) & (v >= -eps) & (u + v <= 1 + eps)
    ok &= (t >= t_min) & (t <= t_max)
    t = np.where(ok, t, np.inf)
    best = t.min(axis=1)
    cands = []
    for i in range(len(rays)):
        if not np.isfinite(best[i]):
            cands.append(set())
        else:
            cands.append(set(np.nonzero(np.abs(t[i] - best[i]) <= rel_tie * max(1.0, abs(best[i])))[0].tolist()))
    return best, cands


def encode_png(samples, ctype, depth=8, palette=None, interlace=False, filters=(0, 1, 2, 3, 4)):
    """Minimal PNG writer for the decoder tests: `samples` is (H, W, channels) of sample values below
    2**depth; rows cycle through the given filter types; `interlace` writes the seven Adam7 passes."""
    import struct
    import zlib

    samples = np.asarray(samples)
    h, w, ch = samples.shape
    bpp = max(1, ch * depth // 8)

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)

    def rows_of(sub):
        if depth == 8:
            return sub.reshape(sub.shape[0], -1).astype(np.uint8)
        if depth == 16:
            return sub.astype(">u2").reshape(sub.shape[0], -1).view(np.uint8)
        bits = ((sub[..., 0][..., None].astype(np.int64) >> np.arange(depth - 1, -1, -1)) & 1).astype(np.uint8)
        return np.packbits(bits.reshape(sub.shape[0], -1), axis=1)

    def filtered(rows, first_filter):
        out = bytearray()
        prev = np.zeros(rows.shape[1], dtype=np.int32)
        for y, row in enumerate(rows.astype(np.int32)):
            f = filters[(first_filter + y) % len(filters)]
            left = np.concatenate([np.zeros(bpp, np.int32), row[:-bpp]]) if rows.shape[1] > bpp else np.zeros_like(row)
            upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if rows.shape[1] > bpp else np.zeros_like(row)
            if f == 0:
                pred = np.zeros_like(row)
            elif f == 1:
                pred = left
            elif f == 2:
                pred = prev
            elif f == 3:
                pred = (left + prev) >> 1
            else:
                p = left + prev - upleft
                pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
                pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            out.append(f)
            out += ((row - pred) & 0xFF).astype(np.uint8).tobytes()
            prev = row
        return bytes(out)

    raw = b""
    passes = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)) if interlace else ((0, 0, 1, 1),)
    for k, (x0, y0, dx, dy) in enumerate(passes):
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] and sub.shape[1]:
            raw += filtered(rows_of(sub), k)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if palette is not None:
        png += chunk(b"PLTE", np.asarray(palette, dtype=np.uint8).tobytes())
    data = zlib.compress(raw, 6)
    half = len(data) // 2
    png += chunk(b"tEXt", b"Comment\0two IDAT chunks follow") + chunk(b"IDAT", data[:half]) + chunk(b"IDAT", data[half:])
    return png + chunk(b"IEND", b"")


def random_scene(seed):
    """A small random scene for differential testing: meshes of random triangles (some with vertex normals and
    texture coordinates, some coplanar and axis-aligned so the 0.001 padding and exact ties occur, some
    duplicated or degenerate), random spheres, and every material and texture kind.  Returns the scene and
    camera settings looking at it."""
    from raytracer_amd import SceneDescription, Sphere
    from raytracer_amd.scene import CameraSettings, Texture, TriangleMesh

    rng = np.random.default_rng(seed)
    sc = SceneDescription(spheres_first=bool(rng.integers(0, 2)))
    f32 = np.float32

    def colour():
        return tuple(rng.random(3).astype(f32))

    def material(depth=0):
        kind = rng.integers(0, 7 if depth < 3 else 5)
        if kind == 0:
            return sc.lambertian(colour())
        if kind == 1:
            return sc.metal(colour(), rng.random() * 0.6)
        if kind == 2:
            return sc.dielectric(1.0 + rng.random())
        if kind == 3:
            return sc.diffuse_light(colour(), 1.0 + 10 * rng.random())
        if kind == 4:
            even = sc.add_texture(Texture.solid_color(colour()))
            odd = sc.add_texture(Texture.solid_color(colour()))
            if rng.random() < 0.5:
                tex = sc.add_texture(Texture.checkerboard(1.0 / (0.05 + rng.random()), even, odd))
            else:
                tex = sc.add_texture(Texture.from_image(rng.integers(0, 256, (int(rng.integers(1, 9)), int(rng.integers(1, 9)), 3))))
            return sc.lambertian_texture(tex)
        return sc.mix(material(depth + 1), material(depth + 1), float(rng.choice([0.0, 1.0, 0.25, 0.5, rng.random()])))

    for _ in range(int(rng.integers(1, 6))):
        nf = int(rng.integers(1, 60))
        style = rng.integers(0, 4)
        if style == 0:  # free triangles
            pos = (rng.random((nf * 3, 3)) * 4 - 2).astype(f32)
        elif style == 1:  # small triangles
            c = rng.random((nf, 1, 3)) * 4 - 2
            pos = (c + (rng.random((nf, 3, 3)) - 0.5) * 0.4).reshape(-1, 3).astype(f32)
        elif style == 2:  # an axis-aligned plane cut into triangles that share edges and vertices exactly
            axis = int(rng.integers(0, 3))
            g = int(rng.integers(2, 6))
            u, v = np.meshgrid(np.linspace(-2, 2, g), np.linspace(-2, 2, g), indexing="ij")
            grid = np.zeros((g, g, 3))
            grid[..., (axis + 1) % 3] = u
            grid[..., (axis + 2) % 3] = v
            grid[..., axis] = rng.choice([-2.0, 0.0, 1.5])
            tris = []
            for i in range(g - 1):
                for j in range(g - 1):
                    a, b, c, d = grid[i, j], grid[i + 1, j], grid[i + 1, j + 1], grid[i, j + 1]
                    tris += [a, b, c, a, c, d]
            pos = np.array(tris, dtype=f32)
            nf = len(pos) // 3
        else:  # duplicates and degenerate triangles
            base = (rng.random((3, 3)) * 3 - 1.5).astype(f32)
            pos = np.concatenate([base, base, base[[0, 0, 1]], base[[0, 1, 1]] * f32(1.0), (rng.random((3, 3)) * 3 - 1.5).astype(f32)])
            nf = len(pos) // 3
        idx = np.arange(nf * 3, dtype=np.uint32).reshape(nf, 3)
        with_attrs = rng.random() < 0.5
        normals = rng.normal(size=(nf * 3, 3)).astype(f32) if with_attrs else np.zeros((0, 3), f32)
        uvs = (rng.random((nf * 3, 2)) * 1.5 - 0.25).astype(f32) if with_attrs else np.zeros((0, 2), f32)
        sc.meshes.append(TriangleMesh(pos, idx, normals, uvs, material()))
    for _ in range(int(rng.integers(0, 12))):
        sc.spheres.append(Sphere(tuple((rng.random(3) * 4 - 2).astype(f32)), float(f32(0.05 + rng.random() * 0.8)), material()))
    eye = rng.normal(size=3)
    eye = eye / np.linalg.norm(eye) * (4.0 + 3.0 * rng.random())
    cs = CameraSettings.look_at(tuple(eye), (0.0, 0.0, 0.0), 40.0 + 30.0 * rng.random(), focus_dist=float(np.linalg.norm(eye)),
                                defocus_angle=float(rng.choice([0.0, 0.0, 2.0])), z_near=1e-3)
    return sc, cs


# ---------------------------------------------------------------------------------------------------------
# Converged-image tolerance: an image `g` rendered with one random stream against two renders `a`, `b` of the
# same scene with the reference's generator (per-thread xoshiro256++) under two different seeds.  All three are
# means over the same number of samples, shape (H, W, 3).
#   sigma_mc   per channel, the Monte-Carlo standard deviation of a pixel's mean: RMSE(a, b) / sqrt(2) over the
#              pixels finite in all three images
#   rmse       per channel RMSE(g, a) and RMSE(g, b) over the same pixels must be <= 3 sigma_mc (two independent
#              unbiased renders differ by sqrt(2) sigma_mc), and their ratio to RMSE(a, b) must lie in
#              [0.75, 1.33] (same per-pixel variance).  Before the RMSEs are taken, values are clamped to the 99.9th
#              percentile of the two reference images' finite values: the per-sample square root leaves a heavy tail
#              (a handful of pixels that caught the light through a rare path), and one such pixel present in one
#              image and absent from the other two would otherwise decide the whole-image RMSE (the oracle's
#              per-thread generator hands work to threads dynamically, so a and b differ from run to run)
#   luminance  mean Rec. 709 luminance over each image's finite pixels: |lum(g) - lum(a or b)| <= the larger of
#              1 % and four standard errors of that difference (from the per-pixel variance)
#   inf        fraction of pixels holding an infinity: within 20 % (relative) of a's and b's, plus 2e-4 absolute
# ---------------------------------------------------------------------------------------------------------
def image_tolerance_report(g, a, b):
    g, a, b = (np.asarray(x, dtype=np.float64) for x in (g, a, b))
    fin = (np.isfinite(g) & np.isfinite(a) & np.isfinite(b)).all(axis=2)
    rep = {"finite_pixels": int(fin.sum()), "channels": []}
    for c in range(3):
        cap = float(np.percentile(np.concatenate([a[..., c][fin], b[..., c][fin]]), 99.9))
        rm = lambda x, y: float(np.sqrt(np.mean((np.minimum(x[..., c][fin], cap) - np.minimum(y[..., c][fin], cap)) ** 2)))
        ab, ga, gb = rm(a, b), rm(g, a), rm(g, b)
        rep["channels"].append({"sigma_mc": ab / np.sqrt(2.0), "rmse_ab": ab, "rmse_ga": ga, "rmse_gb": gb})
    w = np.array([0.2126, 0.7152, 0.0722])
    lum = lambda x: (x @ w)
    la, lb, lg = lum(a), lum(b), lum(g)
    mean_fin = lambda l: float(l[np.isfinite(l)].mean())
    rep["lum"] = {"g": mean_fin(lg), "a": mean_fin(la), "b": mean_fin(lb),
                  # standard error of one image's mean luminance, from the per-pixel variance (a - b)^2 / 2
                  "stderr": float(np.sqrt(np.sum((la[fin] - lb[fin]) ** 2 / 2.0)) / max(1, fin.sum()))}
    rep["inf"] = {k: float(np.isinf(x).any(axis=2).mean()) for k, x in (("g", g), ("a", a), ("b", b))}
    return rep


def assert_image_within_tolerance(g, a, b):
    rep = image_tolerance_report(g, a, b)
    assert rep["finite_pixels"] > 0.5 * g.shape[0] * g.shape[1], rep
    for ch in rep["channels"]:
        for k in ("rmse_ga", "rmse_gb"):
            assert ch[k] <= 3.0 * ch["sigma_mc"] + 1e-12, rep
            assert 0.75 * ch["rmse_ab"] - 1e-12 <= ch[k] <= 1.33 * ch["rmse_ab"] + 1e-12, rep
    lum = rep["lum"]
    for k in ("a", "b"):
        tol = max(0.01 * abs(lum[k]), 4.0 * np.sqrt(2.0) * lum["stderr"])
        assert abs(lum["g"] - lum[k]) <= tol, rep
    inf = rep["inf"]
    for k in ("a", "b"):
        assert abs(inf["g"] - inf[k]) <= 0.2 * max(inf["g"], inf[k]) + 2e-4, rep
    return rep
