"""ctypes view of include/rt_mi355x.h and the loader of the HIP library.

The structures mirror the header field for field; `load_library()` opens the in-tree
`raytracer_amd/csrc/librt_mi355x.so` and fails loudly when it is missing — there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_MI355X_LIB") or os.path.join(_HERE, "csrc", "librt_mi355x.so")

RT_OK = 0
RT_ERR_INVALID_ARGUMENT = -1
RT_ERR_UNSUPPORTED = -2
RT_ERR_OUT_OF_MEMORY = -3
RT_ERR_DEVICE = -4
RT_ERR_NO_DEVICE = -5
RT_ERR_IO = -6
RT_ERR_PARSE = -7

RT_TEX_SOLID, RT_TEX_CHECKER, RT_TEX_IMAGE = 0, 1, 2
(RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_DIELECTRIC, RT_MAT_DIFFUSE_LIGHT, RT_MAT_MIX,
 RT_MAT_TROWBRIDGE_REITZ) = range(6)
RT_MODE_LINEAR_MEAN, RT_MODE_GAMMA_PER_SAMPLE = 0, 1
RT_MISS = 0xFFFFFFFF
RT_NOALLOC_MAX_RAYS = 65536
RT_CTX_TIMING, RT_CTX_COUNTERS = 1, 2
(RT_PIPELINE_AUTO, RT_PIPELINE_SPLIT, RT_PIPELINE_FUSED, RT_PIPELINE_FUSED_LDS, RT_PIPELINE_HYBRID,
 RT_PIPELINE_HYBRID_LDS, RT_PIPELINE_SPLIT_LDS, RT_PIPELINE_PATHS, RT_PIPELINE_PATHS_LDS, RT_PIPELINE_LANES) = range(10)

_f3 = C.c_float * 3


class rt_texture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("rgb", _f3), ("inv_scale", C.c_float), ("even", C.c_int32),
                ("odd", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("pixels", C.POINTER(C.c_uint8))]


class rt_material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("texture", C.c_int32), ("param", C.c_float), ("left", C.c_int32),
                ("right", C.c_int32)]


class rt_mesh(C.Structure):
    _fields_ = [("num_vertices", C.c_uint32), ("positions", C.POINTER(C.c_float)),
                ("num_normals", C.c_uint32), ("normals", C.POINTER(C.c_float)),
                ("num_uvs", C.c_uint32), ("uvs", C.POINTER(C.c_float)),
                ("num_faces", C.c_uint32), ("indices", C.POINTER(C.c_uint32)),
                ("material", C.c_int32)]


class rt_sphere(C.Structure):
    _fields_ = [("center", _f3), ("radius", C.c_float), ("material", C.c_int32)]


class rt_scene_desc(C.Structure):
    _fields_ = [("num_meshes", C.c_uint32), ("meshes", C.POINTER(rt_mesh)),
                ("num_spheres", C.c_uint32), ("spheres", C.POINTER(rt_sphere)),
                ("num_materials", C.c_uint32), ("materials", C.POINTER(rt_material)),
                ("num_textures", C.c_uint32), ("textures", C.POINTER(rt_texture)),
                ("spheres_first", C.c_uint32)]


class rt_scene_info(C.Structure):
    _fields_ = [("num_triangles", C.c_uint32), ("num_spheres", C.c_uint32), ("num_nodes", C.c_uint32),
                ("depth", C.c_uint32), ("bounds_min", _f3), ("bounds_max", _f3),
                ("device_bytes", C.c_uint64)]


class rt_camera_settings(C.Structure):
    _fields_ = [("y_fov", C.c_float), ("z_near", C.c_float), ("z_far", C.c_float),
                ("transform", C.c_float * 12), ("focus_dist", C.c_float), ("defocus_angle", C.c_float)]


class rt_camera(C.Structure):
    _fields_ = [("center", _f3), ("pixel_00_loc", _f3), ("pixel_delta_u", _f3), ("pixel_delta_v", _f3),
                ("defocus_disk_u", _f3), ("defocus_disk_v", _f3), ("defocus_angle", C.c_float),
                ("z_near", C.c_float), ("z_far", C.c_float)]


class rt_settings(C.Structure):
    _fields_ = [("image_width", C.c_uint32), ("image_height", C.c_uint32), ("max_depth", C.c_uint32),
                ("samples_per_pixel", C.c_uint32), ("background_color", _f3), ("mode", C.c_uint32),
                ("seed", C.c_uint64), ("sample_begin", C.c_uint32), ("sample_count", C.c_uint32)]


class rt_ray(C.Structure):
    _fields_ = [("origin", _f3), ("direction", _f3)]


class rt_hit(C.Structure):
    _fields_ = [("t", C.c_float), ("prim_id", C.c_uint32), ("u", C.c_float), ("v", C.c_float),
                ("normal", _f3), ("front_facing", C.c_uint32)]


class rt_stats(C.Structure):
    _fields_ = [("rays_traced", C.c_uint64), ("paths", C.c_uint64), ("node_visits", C.c_uint64),
                ("triangle_tests", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("extend_launches", C.c_uint32), ("shade_launches", C.c_uint32),
                ("total_ms", C.c_double), ("generate_ms", C.c_double), ("extend_ms", C.c_double),
                ("shade_ms", C.c_double), ("accumulate_ms", C.c_double),
                ("timed_extend_rays", C.c_uint64), ("timed_extend_launches", C.c_uint32),
                ("pipeline", C.c_uint32), ("bounce_ms", C.c_double), ("timed_bounce_rays", C.c_uint64),
                ("timed_bounce_launches", C.c_uint32), ("bounce_launches", C.c_uint32),
                ("first_bounce_ms", C.c_double), ("first_bounce_rays", C.c_uint64),
                ("first_bounce_launches", C.c_uint32), ("reserved", C.c_uint32)]


# numpy dtypes with the same layout as rt_ray / rt_hit (both are 4-byte-aligned plain records).
RAY_DTYPE = [("origin", "<f4", 3), ("direction", "<f4", 3)]
HIT_DTYPE = [("t", "<f4"), ("prim_id", "<u4"), ("u", "<f4"), ("v", "<f4"), ("normal", "<f4", 3),
             ("front_facing", "<u4")]

# name -> (restype, argtypes); every symbol include/rt_mi355x.h declares.
_vp = C.c_void_p
EXPORTS = {
    "rt_last_error": (C.c_char_p, []),
    "rt_device_count": (C.c_int, []),
    "rt_context_create": (C.c_int, [C.c_int, _vp, C.c_uint32, C.POINTER(_vp)]),
    "rt_context_destroy": (C.c_int, [_vp]),
    "rt_context_synchronize": (C.c_int, [_vp]),
    "rt_context_set_max_paths": (C.c_int, [_vp, C.c_uint64]),
    "rt_context_set_timed_bounces": (C.c_int, [_vp, C.c_uint32]),
    "rt_context_set_pipeline": (C.c_int, [_vp, C.c_uint32]),
    "rt_scene_create": (C.c_int, [_vp, C.POINTER(rt_scene_desc), C.POINTER(_vp)]),
    "rt_scene_destroy": (C.c_int, [_vp]),
    "rt_scene_get_info": (C.c_int, [_vp, C.POINTER(rt_scene_info)]),
    "rt_scene_export_bvh": (C.c_int, [_vp, _vp, _vp, _vp]),
    "rt_camera_settings_default": (C.c_int, [C.POINTER(rt_camera_settings)]),
    "rt_camera_from_settings": (C.c_int, [C.POINTER(rt_camera_settings), C.c_uint32, C.c_uint32,
                                          C.POINTER(rt_camera)]),
    "rt_trace_closest": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_float, C.c_float, C.c_int, _vp,
                                   C.POINTER(rt_stats)]),
    "rt_trace_closest_device": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_float, C.c_float, C.c_int, _vp,
                                          C.POINTER(rt_stats)]),
    "rt_trace_closest_noalloc": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_float, C.c_float, C.c_int, _vp]),
    "rt_render": (C.c_int, [_vp, _vp, C.POINTER(rt_camera), C.POINTER(rt_settings), _vp,
                            C.POINTER(rt_stats)]),
    "rt_gltf_load": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "rt_gltf_free": (C.c_int, [_vp]),
    "rt_gltf_scene": (C.POINTER(rt_scene_desc), [_vp]),
    "rt_gltf_camera_count": (C.c_uint32, [_vp]),
    "rt_gltf_camera": (C.c_int, [_vp, C.c_uint32, C.POINTER(rt_camera_settings)]),
    "rt_image_decode": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                  C.POINTER(C.POINTER(C.c_uint8))]),
    "rt_image_free": (None, [C.POINTER(C.c_uint8)]),
    "rt_render_device": (C.c_int, [_vp, _vp, C.POINTER(rt_camera), C.POINTER(rt_settings), _vp,
                                   C.POINTER(rt_stats)]),
    "rt_accum_create": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.POINTER(_vp)]),
    "rt_accum_download": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "rt_accum_destroy": (C.c_int, [_vp, _vp]),
    "rt_allreduce_sum": (C.c_int, [C.POINTER(_vp), C.c_int, C.POINTER(_vp), C.c_size_t]),
    "rt_render_node": (C.c_int, [C.POINTER(_vp), C.POINTER(_vp), C.c_int, C.POINTER(rt_camera),
                                 C.POINTER(rt_settings), _vp]),
}


class RtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"rt error {code}: {message}")
        self.code = code
        self.message = message


_lib = None


def load_library(path: str | None = None) -> C.CDLL:
    """Open librt_mi355x.so and bind every declared entry point.  Raises if the library is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # PyTorch-ROCm wheels bundle their own HIP runtime (libamdhip64.so.7).  Two HIP runtimes in one process
    # cannot both own the GPU, so when torch is installed it is imported FIRST: the dynamic loader then
    # satisfies this library's libamdhip64.so.7 dependency with the copy torch already mapped (same SONAME)
    # and torch tensors, streams and RCCL share one runtime with the renderer.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise ImportError(
            f"{p} is missing: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()')"
            " — there is no CPU fallback for the render path")
    lib = C.CDLL(p)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def check(lib: C.CDLL, code: int) -> None:
    if code != RT_OK:
        msg = lib.rt_last_error()
        raise RtError(code, msg.decode("utf-8", "replace") if msg else "")
