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
