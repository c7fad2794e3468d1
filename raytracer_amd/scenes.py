"""Procedural scenes for the benchmark configurations that are not glTF files.

The reference has no scene builders besides its glTF loader; these assemble scenes out of the same object
kinds (`Sphere::new`, `TriangleMesh::new`, `Material::{lambertian, metal, dielectric}`,
src/object/sphere.rs:21, src/object/triangle_mesh.rs:68, src/material.rs:151-175) with a fixed generator so
every run and every rank builds the identical scene.
"""
from __future__ import annotations

import math

import numpy as np

from . import _ffi as F
from .scene import CameraSettings, SceneDescription, Sphere, TriangleMesh

f32 = np.float32
_MASK = (1 << 64) - 1


class SplitMix64:
    """splitmix64; `unit()` takes the top 24 bits as a binary32 in [0, 1)."""

    def __init__(self, seed: int):
        self.state = seed & _MASK

    def next_u64(self) -> int:
        self.state = (self.state + 0x9E3779B97F4A7C15) & _MASK
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
        return z ^ (z >> 31)

    def unit(self) -> float:
        return float(f32(self.next_u64() >> 40) * f32(1.0 / 16777216.0))

    def range(self, lo: float, hi: float) -> float:
        return float(f32(lo) + f32(hi - lo) * f32(self.unit()))


def splitmix64_units(seed: int, first: int, count: int) -> np.ndarray:
    """Values first .. first + count - 1 of SplitMix64(seed)'s `unit()` stream (value k is a function of seed and k
    alone: the state after k + 1 increments, finalised), computed array-wise.  Depends on nothing but 64-bit integer
    arithmetic, so the scenes built from it are the same on every machine and under every numpy."""
    k = np.arange(first + 1, first + count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & _MASK) + k * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return (z >> np.uint64(40)).astype(np.float32) * f32(1.0 / 16777216.0)


def rtiow_final_scene(seed: int = 0x5EED):
    """The "final scene" of Ray Tracing in One Weekend: a ground sphere, a 22 x 22 grid of small random
    spheres (80 % lambertian, 15 % metal, 5 % glass) and three large ones.  Returns (scene, camera settings).
    """
    rng = SplitMix64(seed)
    sc = SceneDescription()
    ground = sc.lambertian((0.5, 0.5, 0.5))
    sc.spheres.append(Sphere((0.0, -1000.0, 0.0), 1000.0, ground))
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose_mat = rng.unit()
            cx = float(f32(a) + f32(0.9) * f32(rng.unit()))
            cz = float(f32(b) + f32(0.9) * f32(rng.unit()))
            center = (cx, 0.2, cz)
            if math.sqrt((cx - 4.0) ** 2 + (0.2 - 0.2) ** 2 + cz ** 2) <= 0.9:
                continue
            if choose_mat < 0.8:
                albedo = tuple(rng.unit() * rng.unit() for _ in range(3))
                m = sc.lambertian(albedo)
            elif choose_mat < 0.95:
                albedo = tuple(rng.range(0.5, 1.0) for _ in range(3))
                m = sc.metal(albedo, rng.range(0.0, 0.5))
            else:
                m = sc.dielectric(1.5)
            sc.spheres.append(Sphere(center, 0.2, m))
    sc.spheres.append(Sphere((0.0, 1.0, 0.0), 1.0, sc.dielectric(1.5)))
    sc.spheres.append(Sphere((-4.0, 1.0, 0.0), 1.0, sc.lambertian((0.4, 0.2, 0.1))))
    sc.spheres.append(Sphere((4.0, 1.0, 0.0), 1.0, sc.metal((0.7, 0.6, 0.5), 0.0)))
    cam = CameraSettings.look_at((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0, focus_dist=10.0, defocus_angle=0.6,
                                 z_near=0.001)
    sc.cameras.append(cam)
    return sc, cam


def triangle_soup(n_triangles: int = 10_000_000, seed: int = 0xC0FFEE, emissive_fraction: float = 0.001):
    """`n_triangles` small triangles with centroids uniform in [-1, 1]^3 and vertices within +-0.005 of the
    centroid; Lambertian albedo 0.8 except a fraction of emitters of strength 50.  No normals and no
    texture coordinates, so every hit takes the geometric-normal path.  Generator: SplitMix64(seed), twelve values per
    triangle, so the first k triangles do not depend on how many follow.  Returns (scene, camera settings)."""
    sc = SceneDescription()
    n_emit = max(1, int(round(n_triangles * emissive_fraction))) if emissive_fraction > 0 else 0
    n_diff = n_triangles - n_emit
    diffuse = sc.lambertian((0.8, 0.8, 0.8))
    emit = sc.diffuse_light((1.0, 1.0, 1.0), 50.0)
    drawn = [0]  # triangles generated so far: triangle t takes values 12 t .. 12 t + 11 of the SplitMix64(seed) stream

    def mesh(n, material):
        chunk = 1 << 20
        pos = np.empty((n * 3, 3), dtype=f32)
        for s in range(0, n, chunk):
            e = min(n, s + chunk)
            u = splitmix64_units(seed, 12 * (drawn[0] + s), 12 * (e - s)).reshape(e - s, 4, 3)
            c = u[:, :1, :] * f32(2.0) - f32(1.0)          # centroid, uniform in [-1, 1]^3
            off = u[:, 1:, :] * f32(0.01) - f32(0.005)     # the three vertices, within +-0.005 of it
            pos[3 * s:3 * e] = (c + off).reshape(-1, 3)
        drawn[0] += n
        idx = np.arange(n * 3, dtype=np.uint32).reshape(n, 3)
        return TriangleMesh(pos, idx, np.zeros((0, 3), f32), np.zeros((0, 2), f32), material)

    if n_diff:
        sc.meshes.append(mesh(n_diff, diffuse))
    if n_emit:
        sc.meshes.append(mesh(n_emit, emit))
    tr = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 3.7833845615386963], dtype=f32)
    cam = CameraSettings(None, float(f32(0.691111147403717)), float(f32(0.1)), 1000.0, tr, 10.0, 0.0)
    sc.cameras.append(cam)
    return sc, cam


def scene_digest(sc: SceneDescription, max_triangles: int | None = None) -> str:
    """sha256 over everything a render of `sc` depends on: vertex positions, indices, normals, texture coordinates and
    material of every mesh (only the first `max_triangles` triangles of the scene when given), spheres, materials,
    textures (image texels included).  Floats enter as their binary32 bytes."""
    import hashlib

    h = hashlib.sha256()

    def put(a, dtype):
        h.update(np.ascontiguousarray(np.asarray(a, dtype=dtype)).tobytes())

    left = max_triangles
    for m in sc.meshes:
        n = len(m.indices) if left is None else min(left, len(m.indices))
        idx = np.asarray(m.indices[:n], dtype=np.uint32)
        put(idx, np.uint32)
        if left is None:
            put(m.positions, f32), put(m.normals, f32), put(m.uvs, f32)
        else:  # the vertices those triangles name, in the order they name them
            put(np.asarray(m.positions, f32)[idx.reshape(-1)], f32)
            left -= n
        put([m.material], np.int32)
        if left == 0:
            break
    for sp in sc.spheres:
        put(list(sp.center) + [sp.radius], f32), put([sp.material], np.int32)
    for m in sc.materials:
        put([m.kind, m.texture, m.left, m.right], np.int32), put([m.param], f32)
    for t in sc.textures:
        put([t.kind, t.even, t.odd], np.int32), put(list(t.rgb) + [t.inv_scale], f32)
        if t.image is not None:
            put(t.image.shape, np.int64), put(t.image, np.uint8)
    return h.hexdigest()


def lit_teapot(teapot: SceneDescription):
    """The reference's teapot asset has neither a light nor a camera, so its render is black.  This variant
    adds a 30 x 30 emitter of strength 200 above it and a camera at (0, 10, 60) looking at the origin."""
    sc = SceneDescription(list(teapot.meshes), list(teapot.spheres), list(teapot.materials), list(teapot.textures))
    light = sc.diffuse_light((1.0, 1.0, 1.0), 200.0)
    pos = np.array([[-15, 30, -15], [15, 30, -15], [15, 30, 15], [-15, 30, 15]], dtype=f32)
    idx = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32)
    sc.meshes.append(TriangleMesh(pos, idx, np.zeros((0, 3), f32), np.zeros((0, 2), f32), light))
    cam = CameraSettings.look_at((0.0, 10.0, 60.0), (0.0, 0.0, 0.0), 40.0, z_near=0.1)
    sc.cameras.append(cam)
    return sc, cam


def cornell_with(scene: SceneDescription, kinds=("metal", "dielectric", "checker")):
    """Cornell box variant that swaps the two boxes' materials so Metal, Dielectric and Checkerboard code
    paths are exercised on triangle geometry.  Mesh 4 is the big box, mesh 5 the small one."""
    sc = SceneDescription(list(scene.meshes), [], list(scene.materials), list(scene.textures), list(scene.cameras))
    meshes = list(sc.meshes)
    if "metal" in kinds:
        m = sc.metal((0.8, 0.85, 0.9), 0.1)
        meshes[4] = TriangleMesh(meshes[4].positions, meshes[4].indices, meshes[4].normals, meshes[4].uvs, m)
    if "dielectric" in kinds:
        m = sc.dielectric(1.5)
        meshes[5] = TriangleMesh(meshes[5].positions, meshes[5].indices, meshes[5].normals, meshes[5].uvs, m)
    if "checker" in kinds:
        from .scene import Texture
        even = sc.add_texture(Texture.solid_color((0.9, 0.9, 0.9)))
        odd = sc.add_texture(Texture.solid_color((0.2, 0.3, 0.1)))
        chk = sc.add_texture(Texture.checkerboard(4.0, even, odd))
        m = sc.lambertian_texture(chk)
        meshes[0] = TriangleMesh(meshes[0].positions, meshes[0].indices, meshes[0].normals, meshes[0].uvs, m)
    sc.meshes = meshes
    return sc
