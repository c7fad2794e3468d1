"""Host-side mirror of the reference's `Renderer` (src/renderer.rs:18-157) over the C ABI.

`Context`, `Scene` wrap the opaque handles; `Renderer` keeps the reference's constructor and its two entry
points, `render` and `render_progressive`.  All rendering work happens inside librt_mi355x.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import numpy as np

from . import _ffi as F
from .scene import CameraSettings, RenderSettings, SceneDescription


class Context:
    """One GPU + one stream (rt_context)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None, flags: int = 0):
        self.handle = None
        self.lib = F.load_library()
        h = C.c_void_p()
        F.check(self.lib, self.lib.rt_context_create(device, C.c_void_p(stream or 0), flags, C.byref(h)))
        self.handle = h
        self.device = device

    def synchronize(self) -> None:
        F.check(self.lib, self.lib.rt_context_synchronize(self.handle))

    def set_max_paths(self, n: int) -> None:
        F.check(self.lib, self.lib.rt_context_set_max_paths(self.handle, n))

    def set_pipeline(self, pipeline: int) -> None:
        F.check(self.lib, self.lib.rt_context_set_pipeline(self.handle, pipeline))

    def set_timed_bounces(self, n: int) -> None:
        F.check(self.lib, self.lib.rt_context_set_timed_bounces(self.handle, n))

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.rt_context_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    """A scene after `build_bvh` (src/scene.rs:29-49), resident in HBM (rt_scene)."""

    def __init__(self, ctx: Context, description: SceneDescription):
        self.handle = None
        self.ctx = ctx
        self.lib = ctx.lib
        desc, keep = description.to_desc()
        h = C.c_void_p()
        F.check(self.lib, self.lib.rt_scene_create(ctx.handle, C.byref(desc), C.byref(h)))
        del keep
        self.handle = h
        self.description = description

    def info(self) -> F.rt_scene_info:
        i = F.rt_scene_info()
        F.check(self.lib, self.lib.rt_scene_get_info(self.handle, C.byref(i)))
        return i

    def export_bvh(self):
        n = self.info().num_nodes
        boxes = np.zeros((n, 6), dtype=np.float32)
        prim = np.zeros(n, dtype=np.int32)
        skip = np.zeros(n, dtype=np.uint32)
        F.check(self.lib, self.lib.rt_scene_export_bvh(self.handle, boxes.ctypes.data, prim.ctypes.data,
                                                        skip.ctypes.data))
        return boxes, prim, skip

    def trace_closest(self, rays: np.ndarray, t_min: float, t_max: float, use_bvh: bool = True):
        """`Hittable::hit` on the root for each ray; returns (hits, stats)."""
        rays = np.ascontiguousarray(rays, dtype=F.RAY_DTYPE)
        out = np.zeros(len(rays), dtype=F.HIT_DTYPE)
        st = F.rt_stats()
        F.check(self.lib, self.lib.rt_trace_closest(self.ctx.handle, self.handle, rays.ctypes.data, len(rays),
                                                     t_min, t_max, 1 if use_bvh else 0, out.ctypes.data,
                                                     C.byref(st)))
        return out, st

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.rt_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def camera_from_settings(cs: CameraSettings, width: int, height: int) -> F.rt_camera:
    """`Camera::new` (src/camera.rs:20-64)."""
    lib = F.load_library()
    cam = F.rt_camera()
    s = cs.to_c()
    F.check(lib, lib.rt_camera_from_settings(C.byref(s), width, height, C.byref(cam)))
    return cam


def should_save_image(current_sample: int, sample_count: int) -> bool:
    """src/renderer.rs:167-171."""
    return current_sample <= 5 or current_sample % 16 == 0 or current_sample == sample_count


def pixels_to_rgb8(pixels: np.ndarray) -> np.ndarray:
    """`pixels_to_image` (src/renderer.rs:173-176): Rgb32F -> Rgb8 as the `image` crate converts it,
    round(clamp(c, 0, 1) * 255); NaN becomes 0."""
    p = np.nan_to_num(np.asarray(pixels, dtype=np.float32), nan=0.0, posinf=1.0, neginf=0.0)
    # Rust's f32::round rounds halves away from zero (np.rint would round them to even).  The product is binary32 as in
    # the reference; the half is added in binary64, where the sum is exact (in binary32 the largest value below 0.5
    # plus 0.5 rounds up to 1).
    scaled = np.clip(p, 0.0, 1.0) * np.float32(255.0)
    return np.floor(scaled.astype(np.float64) + 0.5).astype(np.uint8)


class Renderer:
    """`Renderer::new(camera, scene, render)` (src/renderer.rs:25-31).

    `camera` is an rt_camera (from `camera_from_settings`), `scene` a device-resident `Scene`."""

    def __init__(self, camera: F.rt_camera, scene: Scene, render: RenderSettings):
        self.camera = camera
        self.scene = scene
        self.render_settings = render
        self.last_stats: Optional[F.rt_stats] = None

    def _sums(self, settings: RenderSettings, want_stats: bool = True) -> np.ndarray:
        s = settings.to_c()
        out = np.zeros((settings.image_height, settings.image_width, 3), dtype=np.float32)
        st = F.rt_stats()
        lib = self.scene.lib
        F.check(lib, lib.rt_render(self.scene.ctx.handle, self.scene.handle, C.byref(self.camera), C.byref(s),
                                   out.ctypes.data, C.byref(st) if want_stats else None))
        self.last_stats = st
        return out

    def render_sums(self, mode: Optional[int] = None) -> np.ndarray:
        """Un-normalised per-pixel sums of the configured sample range, shape (H, W, 3)."""
        rs = self.render_settings
        if mode is not None:
            rs = RenderSettings(**{**rs.__dict__, "mode": mode})
        return self._sums(rs)

    def render(self, square_size: int = 16) -> np.ndarray:
        """`Renderer::render` (src/renderer.rs:72-103): linear mean of all samples as an RGB8 image.
        `square_size` only shaped the reference's CPU work chunks and has no effect here."""
        rs = RenderSettings(**{**self.render_settings.__dict__, "mode": F.RT_MODE_LINEAR_MEAN})
        n = rs.sample_count or rs.samples_per_pixel
        sums = self._sums(rs)
        sample_scale = np.float32(1.0) / np.float32(n)
        return pixels_to_rgb8(sums * sample_scale)

    def render_progressive(self, output: Callable[[np.ndarray, int, int], None], square_size: int = 16) -> None:
        """`Renderer::render_progressive` (src/renderer.rs:105-156): the running sum of gamma-mapped samples;
        `output(pixels, width, height)` receives `aggregate / current_sample` after samples 1-5, every 16th
        and the last (src/renderer.rs:138-148,167-171).  Samples between two reports are rendered in one
        device call; the sums are identical to adding one sample at a time."""
        rs0 = self.render_settings
        w, h, total = rs0.image_width, rs0.image_height, rs0.samples_per_pixel
        lib = self.scene.lib
        ctx = self.scene.ctx
        import torch  # device memory for the running aggregate

        agg = torch.zeros(h * w * 3, dtype=torch.float32, device=f"cuda:{ctx.device}")
        torch.cuda.synchronize(ctx.device)
        done = 0
        while done < total:
            nxt = done + 1
            while not should_save_image(nxt, total):
                nxt += 1
            rs = RenderSettings(**{**rs0.__dict__, "mode": F.RT_MODE_GAMMA_PER_SAMPLE, "sample_begin": done,
                                   "sample_count": nxt - done})
            s = rs.to_c()
            F.check(lib, lib.rt_render_device(ctx.handle, self.scene.handle, C.byref(self.camera), C.byref(s),
                                              C.c_void_p(agg.data_ptr()), None))
            ctx.synchronize()
            done = nxt
            sample_scale = np.float32(1.0) / np.float32(done)
            output((agg.cpu().numpy().reshape(h, w, 3) * sample_scale), w, h)
