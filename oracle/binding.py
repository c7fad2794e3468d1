"""ctypes binding of the CPU oracle (oracle/librt_oracle.so).

TEST INFRASTRUCTURE ONLY: import this from tests, the smoke check and bench.py's cpu_baseline leg, never
from the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd import _ffi as F

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_oracle.so")
RNG_COUNTER, RNG_THREAD = 0, 1

_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(
            os.path.join(_HERE, "rt_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "librt_oracle.so"])
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        l = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        l.oracle_last_error.restype = C.c_char_p
        l.oracle_scene_create.argtypes = [C.POINTER(F.rt_scene_desc), C.POINTER(vp)]
        l.oracle_scene_destroy.argtypes = [vp]
        l.oracle_scene_destroy.restype = None
        l.oracle_scene_get_info.argtypes = [vp, C.POINTER(F.rt_scene_info)]
        l.oracle_scene_export_bvh.argtypes = [vp, vp, vp, vp]
        l.oracle_camera_settings_default.argtypes = [C.POINTER(F.rt_camera_settings)]
        l.oracle_camera_from_settings.argtypes = [C.POINTER(F.rt_camera_settings), C.c_uint32, C.c_uint32,
                                                  C.POINTER(F.rt_camera)]
        l.oracle_render.argtypes = [vp, C.POINTER(F.rt_camera), C.POINTER(F.rt_settings), vp,
                                    C.POINTER(F.rt_stats), C.c_int, C.c_int, C.c_int]
        l.oracle_trace_closest.argtypes = [vp, vp, C.c_size_t, C.c_float, C.c_float, C.c_int, vp,
                                           C.POINTER(F.rt_stats)]
        l.oracle_rng_draws.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, vp]
        l.oracle_rng_draws.restype = None
        _lib = l
    return _lib


def _check(code: int) -> None:
    if code != 0:
        raise F.RtError(code, lib().oracle_last_error().decode())


class OracleScene:
    def __init__(self, scene):
        self.handle = None
        d, self._keep = scene.to_desc()
        h = C.c_void_p()
        _check(lib().oracle_scene_create(C.byref(d), C.byref(h)))
        self.handle = h

    def close(self):
        if self.handle:
            lib().oracle_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    def info(self) -> F.rt_scene_info:
        i = F.rt_scene_info()
        _check(lib().oracle_scene_get_info(self.handle, C.byref(i)))
        return i

    def export_bvh(self):
        n = self.info().num_nodes
        boxes = np.zeros((n, 6), dtype=np.float32)
        prim = np.zeros(n, dtype=np.int32)
        skip = np.zeros(n, dtype=np.uint32)
        _check(lib().oracle_scene_export_bvh(self.handle, boxes.ctypes.data, prim.ctypes.data, skip.ctypes.data))
        return boxes, prim, skip

    def trace_closest(self, rays: np.ndarray, t_min: float, t_max: float, use_bvh: bool = True):
        rays = np.ascontiguousarray(rays, dtype=F.RAY_DTYPE)
        out = np.zeros(len(rays), dtype=F.HIT_DTYPE)
        st = F.rt_stats()
        _check(lib().oracle_trace_closest(self.handle, rays.ctypes.data, len(rays), t_min, t_max,
                                          1 if use_bvh else 0, out.ctypes.data, C.byref(st)))
        return out, st

    def render(self, camera: F.rt_camera, settings, threads: int = 8, rng_mode: int = RNG_COUNTER,
               use_bvh: bool = True, accum: np.ndarray | None = None):
        """Returns (sums, stats): un-normalised per-pixel sums, shape (H, W, 3)."""
        s = settings.to_c() if hasattr(settings, "to_c") else settings
        if accum is None:
            accum = np.zeros((s.image_height, s.image_width, 3), dtype=np.float32)
        st = F.rt_stats()
        _check(lib().oracle_render(self.handle, C.byref(camera), C.byref(s), accum.ctypes.data, C.byref(st),
                                   threads, rng_mode, 1 if use_bvh else 0))
        return accum, st


def camera_from_settings(cs, width: int, height: int) -> F.rt_camera:
    c = F.rt_camera()
    s = cs.to_c() if hasattr(cs, "to_c") else cs
    _check(lib().oracle_camera_from_settings(C.byref(s), width, height, C.byref(c)))
    return c


def rng_draws(seed: int, pixel: int, sample: int, n: int) -> np.ndarray:
    out = np.zeros(n, dtype=np.uint32)
    lib().oracle_rng_draws(seed, pixel, sample, n, out.ctypes.data)
    return out
