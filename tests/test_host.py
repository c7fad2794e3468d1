"""CPU tests of the host side: glTF import, the C-ABI surface, camera construction, device math on the host."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import binding as ob
from raytracer_amd import CameraSettings, RenderSettings, _ffi as F
from raytracer_amd.renderer import camera_from_settings, pixels_to_rgb8, should_save_image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gltf_cornell(cornell):
    """assets/cornell.gltf as src/scene.rs:208-250 reads it: 6 mesh nodes, 36 triangles, 1 camera; the light
    becomes a white DiffuseLight of strength 200, everything else Lambertian of the base colour."""
    assert len(cornell.meshes) == 6 and cornell.num_primitives == 36
    assert [len(m.indices) for m in cornell.meshes] == [6, 2, 2, 2, 12, 12]
    kinds = [cornell.materials[m.material].kind for m in cornell.meshes]
    assert kinds == [F.RT_MAT_LAMBERTIAN] * 3 + [F.RT_MAT_DIFFUSE_LIGHT] + [F.RT_MAT_LAMBERTIAN] * 2
    light = cornell.materials[cornell.meshes[3].material]
    assert light.param == 200.0 and tuple(cornell.textures[light.texture].rgb) == (1.0, 1.0, 1.0)
    green = cornell.textures[cornell.materials[cornell.meshes[1].material].texture].rgb
    assert np.allclose(green, (0.0, 0.8003553152084351, 0.000760709575843066))
    # node transform applied to positions (scale then translate), normals untouched
    lp = cornell.meshes[3].positions
    assert np.allclose(lp[:, 1], 0.9856827259063721) and np.allclose(np.abs(lp[:, [0, 2]]), 0.2685281038284302)
    assert np.allclose(np.linalg.norm(cornell.meshes[4].normals, axis=1), 1.0)
    bb = cornell.meshes[4].positions
    assert np.allclose(bb.min(0), [-0.4772908, -1.0027841, -0.4259056], atol=1e-6)
    cam = cornell.cameras[0]
    assert cam.y_fov == pytest.approx(0.691111147403717) and cam.z_far == 1000.0
    assert cam.focus_dist == 10.0 and cam.defocus_angle == 0.0
    assert np.allclose(cam.transform[9:], [0, 0, 3.7833845615386963])
    assert all(len(m.uvs) == len(m.positions) for m in cornell.meshes)


def test_gltf_teapot(teapot):
    """assets/teapot.gltf: base64 buffer, u16 indices, normals but no uvs, node scale 10, no camera."""
    assert len(teapot.meshes) == 1 and teapot.num_primitives == 9128
    m = teapot.meshes[0]
    assert m.positions.shape == (4690, 3) and m.normals.shape == (4690, 3) and m.uvs.shape == (0, 2)
    assert np.allclose(m.positions.max(0), [33.942089, 14.2828, 20.0], atol=1e-4)
    assert teapot.cameras == []
    d = teapot.camera(0)  # falls back to CameraSettings::default (src/scene.rs:25-27,73-85)
    assert d.y_fov == pytest.approx(np.float32(80.0) * np.float32(np.pi / 180)) and np.isinf(d.z_far)
    assert teapot.materials[m.material].kind == F.RT_MAT_LAMBERTIAN
    assert tuple(teapot.textures[teapot.materials[m.material].texture].rgb) == (1.0, 1.0, 1.0)


def test_library_exports_every_declared_symbol():
    """Every function include/rt_mi355x.h declares is exported by the built library and bound by _ffi."""
    text = open(os.path.join(ROOT, "include", "rt_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text))
    assert declared == set(F.EXPORTS), declared ^ set(F.EXPORTS)
    lib = F.load_library()
    for name in declared:
        assert hasattr(lib, name)
    nm = subprocess.run(["nm", "-D", "--defined-only", F.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (rt_[a-z_0-9]+)", nm))
    assert exported == declared, exported ^ declared


def test_struct_layouts_match_the_header():
    """sizeof / offsets as a C compiler sees the header."""
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "rt_mi355x.h"
    int main(void){
      printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(rt_texture), sizeof(rt_material),
        sizeof(rt_mesh), sizeof(rt_sphere), sizeof(rt_scene_desc), sizeof(rt_scene_info), sizeof(rt_camera_settings),
        sizeof(rt_camera), sizeof(rt_settings), sizeof(rt_ray), sizeof(rt_hit), sizeof(rt_stats));
      printf("%zu %zu %zu\n", offsetof(rt_settings, seed), offsetof(rt_stats, total_ms), offsetof(rt_mesh, material));
      return 0; }'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "l.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "l")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True).stdout.split()
    sizes = [C.sizeof(t) for t in (F.rt_texture, F.rt_material, F.rt_mesh, F.rt_sphere, F.rt_scene_desc,
                                   F.rt_scene_info, F.rt_camera_settings, F.rt_camera, F.rt_settings, F.rt_ray,
                                   F.rt_hit, F.rt_stats)]
    assert [int(x) for x in out[:12]] == sizes
    assert [int(x) for x in out[12:]] == [F.rt_settings.seed.offset, F.rt_stats.total_ms.offset,
                                          F.rt_mesh.material.offset]
    assert np.dtype(F.RAY_DTYPE).itemsize == C.sizeof(F.rt_ray)
    assert np.dtype(F.HIT_DTYPE).itemsize == C.sizeof(F.rt_hit)


def test_camera_matches_oracle_and_hand_values(cornell):
    """rt_camera_from_settings (host code of the product) == the oracle's Camera::new, bit for bit, and the
    Cornell camera has the hand-derived geometry: centre (0,0,3.78), |pixel delta| = 2*tan(yfov/2)*10/H."""
    for cs, (w, h) in ((cornell.camera(0), (600, 600)), (cornell.camera(0), (1920, 1080)),
                       (CameraSettings(), (1280, 720)),
                       (CameraSettings.look_at((13, 2, 3), (0, 0, 0), 20.0, defocus_angle=0.6), (400, 225))):
        a = camera_from_settings(cs, w, h)
        b = ob.camera_from_settings(cs, w, h)
        assert bytes(a) == bytes(b)
    cam = camera_from_settings(cornell.camera(0), 600, 600)
    assert list(cam.center) == [0.0, 0.0, np.float32(3.7833845615386963)]
    expect = 2 * np.tan(0.691111147403717 / 2) * 10 / 600
    assert cam.pixel_delta_u[0] == pytest.approx(expect, rel=1e-6) and cam.pixel_delta_v[1] == pytest.approx(-expect, rel=1e-6)
    assert cam.pixel_00_loc[2] == pytest.approx(3.7833845615386963 - 10, rel=1e-6)
    assert cam.defocus_disk_u[0] == 0.0 and cam.z_near == np.float32(0.1)
    lib = F.load_library()
    assert lib.rt_camera_from_settings(None, 1, 1, None) == F.RT_ERR_INVALID_ARGUMENT
    s = cornell.camera(0).to_c()
    out = F.rt_camera()
    assert lib.rt_camera_from_settings(C.byref(s), 0, 10, C.byref(out)) == F.RT_ERR_INVALID_ARGUMENT
    assert b"non-zero" in lib.rt_last_error()


def test_device_sincos_matches_libm_on_host():
    """rt::sin_unit / rt::cos_unit, compiled for the host from the same header the kernels include, equal
    glibc's sinf / cosf bit for bit on [-1, 1] (every 61st binary32 plus the switch points exhaustively; the
    full exhaustive sweep, 2^31 values, also reports zero mismatches)."""
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "sincos_check")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off",
                               os.path.join(ROOT, "tests", "native", "sincos_check.cpp"), "-o", exe, "-lm"])
        out = subprocess.run([exe, "61"], capture_output=True, text=True, check=True).stdout.split()
    checked, bad_s, bad_c = (int(x) for x in out)
    assert checked > 3.4e7 and bad_s == 0 and bad_c == 0


def test_device_acos_atan2_match_libm_on_host():
    """rt::acos_f32 / rt::atan_f32 / rt::atan2_f32 (Sphere::get_uv on the GPU), compiled for the host from the
    same header the kernels include, equal glibc's acosf / atanf / atan2f bit for bit (every 127th binary32 and
    two million argument pairs plus the special values here; the exhaustive sweep of all 2^32 arguments and 10^9
    pairs also reports zero mismatches against glibc 2.35)."""
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "invtrig_check")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off",
                               os.path.join(ROOT, "tests", "native", "invtrig_check.cpp"), "-o", exe, "-lm"])
        out = subprocess.run([exe, "127", "1000000"], capture_output=True, text=True, check=True).stdout.split()
    n1, b1, n2, b2, n3, b3 = (int(x) for x in out)
    assert n1 > 3.3e7 and n3 > 2e6 and (b1, b2, b3) == (0, 0, 0)


def test_random_stream_of_the_kernels_equals_the_oracles_beyond_131072_draws():
    """The per-path random stream of the kernels (rt_math.h, compiled for the host here; the kernels include the
    same header) against the oracle's separate implementation: the first 400000 draws of a few paths, which
    crosses draw 131072 where the block index outgrows the 16 bits the hash key has room for, and the float
    conversions of both."""
    from oracle import binding as ob

    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "rng_dump")
        subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "native", "rng_dump.cpp"), "-o", exe])
        for seed, pixel, sample in ((1, 0, 0), (7, 2073599, 1023), (0xDEADBEEF12345678, (1 << 26) - 1, (1 << 22) - 1)):
            raw = subprocess.run([exe, str(seed), str(pixel), str(sample), "400000"], capture_output=True, check=True).stdout
            mine = np.frombuffer(raw, dtype=np.uint32)
            want = ob.rng_draws(seed, pixel, sample, 400000)
            assert len(mine) == 400000 and np.array_equal(mine, want)
            assert len(np.unique(mine[131072:262144])) > 131000  # the later blocks are not a replay of the first
            assert not np.array_equal(mine[:131072], mine[131072:262144])


def test_product_fails_loudly_without_a_gpu():
    """No CPU fallback: without a HIP device a context cannot be created and the error says so."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = F.load_library()
    h = C.c_void_p()
    rc = lib.rt_context_create(0, None, 0, C.byref(h))
    assert rc == F.RT_ERR_NO_DEVICE and not h.value
    assert b"no CPU path" in lib.rt_last_error() or b"no HIP device" in lib.rt_last_error()
    with pytest.raises(ImportError):
        F.load_library("/nonexistent/librt_mi355x.so")


def test_progressive_schedule_and_rgb8():
    """should_save_image (src/renderer.rs:167-171) and the f32 -> u8 conversion of pixels_to_image."""
    saved = [s for s in range(1, 101) if should_save_image(s, 100)]
    assert saved == [1, 2, 3, 4, 5, 16, 32, 48, 64, 80, 96, 100]
    px = np.array([0.0, 0.5, 1.0, 2.0, -1.0, np.nan, np.inf], dtype=np.float32)
    assert pixels_to_rgb8(px).tolist() == [0, 128, 255, 255, 0, 0, 255]


def test_rgb8_rounds_halves_away_from_zero_in_both_host_layers():
    """`pixels_to_image` (src/renderer.rs:173-176) goes through the `image` crate's f32 -> u8 conversion,
    round(clamp(c, 0, 1) * 255) with Rust's `round` (halves away from zero).  Values whose product with 255 is
    exactly k + 0.5 in binary32 must give k + 1 in the Python sink and in the C++ one (rt_host.hpp)."""
    k = np.arange(255, dtype=np.float32)
    c = ((k + np.float32(0.5)) / np.float32(255.0)).astype(np.float32)
    exact = (c * np.float32(255.0)) == k + np.float32(0.5)
    assert exact.sum() > 20 and (k[exact] % 2 == 0).any()  # even k is where round-half-to-even would differ
    want = (k[exact] + 1).astype(np.uint8)
    assert np.array_equal(pixels_to_rgb8(c[exact]), want)
    # products just below k + 0.5 round down: in binary32 the largest value below 0.5, plus 0.5, would round up to 1
    below = np.nextafter(np.float32(0.5), np.float32(0.0)) / np.float32(255.0)
    assert below * np.float32(255.0) < np.float32(0.5) and pixels_to_rgb8(np.array([below], np.float32))[0] == 0
    c_extra = np.concatenate([c[exact], np.array([below], np.float32)])
    want_extra = np.concatenate([want, np.array([0], np.uint8)])
    src = r"""
#include "raytracer_amd/csrc/host/rt_host.hpp"
int main() { std::vector<float> px; float f; while (fread(&f, 4, 1, stdin) == 1) px.push_back(f);
  std::vector<uint8_t> o = rt::host::pixels_to_rgb8(px); fwrite(o.data(), 1, o.size(), stdout); return 0; }
"""
    with tempfile.TemporaryDirectory() as td:
        cpp, exe = os.path.join(td, "rgb8.cpp"), os.path.join(td, "rgb8")
        open(cpp, "w").write(src)
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", ROOT, cpp, "-o", exe, "-L", os.path.dirname(F.LIB_PATH),
                               "-lrt_mi355x", "-lz", "-Wl,-rpath," + os.path.dirname(F.LIB_PATH)])
        out = subprocess.run([exe], input=c_extra.tobytes(), capture_output=True, check=True).stdout
    assert np.array_equal(np.frombuffer(out, dtype=np.uint8), want_extra)


def test_render_settings_defaults_match_cli():
    """Defaults of the reference CLI (src/main.rs:46-59,86): 1280x720, depth 50, 100 spp, black background."""
    rs = RenderSettings()
    assert (rs.image_width, rs.image_height, rs.max_depth, rs.samples_per_pixel) == (1280, 720, 50, 100)
    assert tuple(rs.background_color) == (0.0, 0.0, 0.0)
    c = rs.to_c()
    assert c.mode == F.RT_MODE_GAMMA_PER_SAMPLE and c.seed == 1


def _same_scene(a, b):
    assert len(a.meshes) == len(b.meshes) and len(a.cameras) == len(b.cameras)
    for ma, mb in zip(a.meshes, b.meshes):
        for fa, fb in ((ma.positions, mb.positions), (ma.normals, mb.normals), (ma.uvs, mb.uvs)):
            assert fa.shape == fb.shape and np.array_equal(fa.view(np.uint32), fb.view(np.uint32))
        assert np.array_equal(ma.indices, mb.indices)
        ka, kb = a.materials[ma.material], b.materials[mb.material]
        assert (ka.kind, ka.param) == (kb.kind, kb.param)
        assert tuple(a.textures[ka.texture].rgb) == tuple(b.textures[kb.texture].rgb)
    for ca, cb in zip(a.cameras, b.cameras):
        assert (ca.y_fov, ca.z_near, ca.z_far, ca.focus_dist, ca.defocus_angle) == (cb.y_fov, cb.z_near, cb.z_far, cb.focus_dist, cb.defocus_angle)
        assert np.array_equal(np.asarray(ca.transform, np.float32), np.asarray(cb.transform, np.float32))


def test_native_gltf_reader_matches_python_reader(cornell, teapot, tmp_path):
    """rt_gltf_load (C++, part of the library) and the Python reader used by the tests produce identical
    scenes, bit for bit, on both reference assets and on a synthetic file with a rotated node, a matrix node,
    u32 indices, an interleaved (strided) vertex buffer, a non-indexed primitive (which yields no faces,
    src/scene.rs:170-175) and a GLB container."""
    import json
    import struct

    from raytracer_amd.scene import load_from_gltf, load_from_gltf_native

    _same_scene(load_from_gltf_native(os.path.join(ROOT, "assets", "cornell.gltf")), cornell)
    _same_scene(load_from_gltf_native(os.path.join(ROOT, "assets", "teapot.gltf")), teapot)

    rng = np.random.default_rng(5)
    verts = rng.normal(size=(6, 3)).astype(np.float32)
    norms = rng.normal(size=(6, 3)).astype(np.float32)
    inter = np.concatenate([verts, norms], axis=1).astype(np.float32).tobytes()  # stride 24
    idx = np.array([0, 1, 2, 3, 4, 5, 1], dtype=np.uint32).tobytes()           # 7 indices: the tail is dropped
    uv = (rng.random((6, 2)) * 65535).astype(np.uint16).tobytes()               # normalised u16 texcoords
    blob = inter + idx + uv
    doc = {
        "asset": {"version": "2.0"},
        "nodes": [
            {"mesh": 0, "translation": [1, 2, 3], "rotation": [0.1825742, 0.3651484, 0.5477226, 0.7302967], "scale": [2, 0.5, 3]},
            {"mesh": 1, "matrix": [0, 1, 0, 0, -1, 0, 0, 0, 0, 0, 1, 0, 5, 6, 7, 1]},
            {"camera": 0, "translation": [0, 1, 9]}, {"camera": 1},
        ],
        "cameras": [{"type": "perspective", "perspective": {"yfov": 0.5, "znear": 0.01}},
                    {"type": "orthographic", "orthographic": {"xmag": 1, "ymag": 1, "zfar": 10, "znear": 1}}],
        "materials": [{"emissiveFactor": [0, 0.5, 0]}, {"pbrMetallicRoughness": {"baseColorFactor": [0.1, 0.2, 0.3, 1]}}],
        "meshes": [
            {"primitives": [{"mode": 1, "attributes": {"POSITION": 0}},
                            {"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 3}, "indices": 2, "material": 0}]},
            {"primitives": [{"attributes": {"POSITION": 0}, "material": 1}]},
        ],
        "accessors": [
            {"bufferView": 0, "componentType": 5126, "count": 6, "type": "VEC3"},
            {"bufferView": 0, "byteOffset": 12, "componentType": 5126, "count": 6, "type": "VEC3"},
            {"bufferView": 1, "componentType": 5125, "count": 7, "type": "SCALAR"},
            {"bufferView": 2, "componentType": 5123, "normalized": True, "count": 6, "type": "VEC2"},
        ],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": len(inter), "byteStride": 24},
                        {"buffer": 0, "byteOffset": len(inter), "byteLength": len(idx)},
                        {"buffer": 0, "byteOffset": len(inter) + len(idx), "byteLength": len(uv)}],
        "buffers": [{"byteLength": len(blob)}],
    }
    text = json.dumps(doc).encode()
    text += b" " * (-len(text) % 4)
    blob += b"\0" * (-len(blob) % 4)
    glb = b"glTF" + struct.pack("<II", 2, 12 + 8 + len(text) + 8 + len(blob))
    glb += struct.pack("<I", len(text)) + b"JSON" + text + struct.pack("<I", len(blob)) + b"BIN\0" + blob
    path = tmp_path / "synthetic.glb"
    path.write_bytes(glb)
    a, b = load_from_gltf_native(str(path)), load_from_gltf(str(path))
    _same_scene(a, b)
    assert [len(m.indices) for m in a.meshes] == [2, 0] and len(a.cameras) == 1
    assert a.materials[a.meshes[0].material].kind == F.RT_MAT_DIFFUSE_LIGHT and a.materials[a.meshes[0].material].param == 1.0
    assert np.isinf(a.cameras[0].z_far)
    # error reporting
    lib = F.load_library()
    h = C.c_void_p()
    assert lib.rt_gltf_load(b"/nonexistent/file.gltf", C.byref(h)) == F.RT_ERR_IO
    bad = tmp_path / "bad.gltf"
    bad.write_text("{ \"nodes\": [ }")
    assert lib.rt_gltf_load(os.fsencode(str(bad)), C.byref(h)) == F.RT_ERR_PARSE and b"JSON" in lib.rt_last_error()


def _textured_gltf(tmp_path, name, png, how):
    """A one-triangle glTF whose material has the PNG as base-colour texture: as data URI, as a file next to
    the document, or in a buffer view of a GLB container."""
    import base64
    import json
    import struct

    verts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32).tobytes()
    uv = np.array([[0, 0], [1, 0], [0, 1]], dtype=np.float32).tobytes()
    idx = np.array([0, 1, 2], dtype=np.uint16).tobytes() + b"\0\0"
    blob = verts + uv + idx
    doc = {
        "asset": {"version": "2.0"}, "nodes": [{"mesh": 0}],
        "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "baseColorFactor": [0.5, 0.5, 0.5, 1]}}],
        "textures": [{"source": 0}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1}, "indices": 2, "material": 0}]}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"},
                      {"bufferView": 1, "componentType": 5126, "count": 3, "type": "VEC2"},
                      {"bufferView": 2, "componentType": 5123, "count": 3, "type": "SCALAR"}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 24},
                        {"buffer": 0, "byteOffset": 60, "byteLength": 6}],
    }
    if how == "glb":
        doc["bufferViews"].append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(png)})
        doc["images"] = [{"bufferView": 3, "mimeType": "image/png"}]
        blob += png
        blob += b"\0" * (-len(blob) % 4)
        doc["buffers"] = [{"byteLength": len(blob)}]
        text = json.dumps(doc).encode()
        text += b" " * (-len(text) % 4)
        data = b"glTF" + struct.pack("<II", 2, 12 + 8 + len(text) + 8 + len(blob))
        data += struct.pack("<I", len(text)) + b"JSON" + text + struct.pack("<I", len(blob)) + b"BIN\0" + blob
        path = tmp_path / f"{name}.glb"
        path.write_bytes(data)
        return str(path)
    doc["buffers"] = [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}]
    if how == "file":
        (tmp_path / f"{name}.png").write_bytes(png)
        doc["images"] = [{"uri": f"{name}.png"}]
    else:
        doc["images"] = [{"uri": "data:image/png;base64," + base64.b64encode(png).decode()}]
    path = tmp_path / f"{name}.gltf"
    path.write_text(json.dumps(doc))
    return str(path)


def test_png_base_colour_textures(tmp_path):
    """Base-colour textures (`load_image`, src/scene.rs:87-114; texel use in src/texture.rs:115-136): the
    library's PNG decoder and the Python reader's give the texels the PNG holds — (l, l, l) for grey, palette
    expanded, alpha dropped, sub-byte grey scaled to 0..255 — for every colour type, filter type, Adam7
    interlacing and all three ways a glTF file can carry an image; 16-bit samples and damaged files are refused.
    Pillow, where importable, cross-checks the test's own PNG writer."""
    import io

    from helpers import encode_png
    from raytracer_amd.png import decode_png_rgb8
    from raytracer_amd.scene import load_from_gltf, load_from_gltf_native

    try:
        from PIL import Image
    except ImportError:
        Image = None
    rng = np.random.default_rng(11)
    h, w = 7, 13
    cases = []
    for il in (False, True):
        rgb = rng.integers(0, 256, (h, w, 3))
        cases.append((f"rgb{il:d}", encode_png(rgb, 2, interlace=il), rgb))
        rgba = rng.integers(0, 256, (h, w, 4))
        cases.append((f"rgba{il:d}", encode_png(rgba, 6, interlace=il), rgba[..., :3]))
        g = rng.integers(0, 256, (h, w, 1))
        cases.append((f"grey{il:d}", encode_png(g, 0, interlace=il), np.repeat(g, 3, 2)))
        ga = rng.integers(0, 256, (h, w, 2))
        cases.append((f"greyalpha{il:d}", encode_png(ga, 4, interlace=il), np.repeat(ga[..., :1], 3, 2)))
        for d, scale in ((1, 255), (2, 85), (4, 17)):
            g = rng.integers(0, 1 << d, (h, w, 1))
            cases.append((f"grey{d}bit{il:d}", encode_png(g, 0, d, interlace=il), np.repeat(g * scale, 3, 2)))
            pal = rng.integers(0, 256, (1 << d, 3))
            cases.append((f"pal{d}bit{il:d}", encode_png(g, 3, d, palette=pal, interlace=il), pal[g[..., 0]]))
        idx = rng.integers(0, 256, (h, w, 1))
        pal = rng.integers(0, 256, (256, 3))
        cases.append((f"pal8{il:d}", encode_png(idx, 3, 8, palette=pal, interlace=il), pal[idx[..., 0]]))
    cases.append(("one_pixel", encode_png(np.array([[[9, 8, 7]]]), 2, interlace=True), np.array([[[9, 8, 7]]])))
    hows = ("uri", "file", "glb")
    for k, (name, png, want) in enumerate(cases):
        want = want.astype(np.uint8)
        assert np.array_equal(decode_png_rgb8(png), want), name
        if Image is not None:
            assert np.array_equal(np.asarray(Image.open(io.BytesIO(png)).convert("RGB")), want), name
        path = _textured_gltf(tmp_path, name, png, hows[k % 3])
        for sc in (load_from_gltf_native(path), load_from_gltf(path)):
            m = sc.materials[sc.meshes[0].material]
            t = sc.textures[m.texture]
            assert m.kind == F.RT_MAT_LAMBERTIAN and t.kind == F.RT_TEX_IMAGE
            assert t.image.shape == want.shape and np.array_equal(t.image, want), name
    if Image is not None:  # a PNG from an independent writer, with its own choice of filters
        big = rng.integers(0, 256, (37, 53, 3)).astype(np.uint8)
        big[5:20] = np.linspace(0, 255, 53).astype(np.uint8)[None, :, None]  # smooth rows make it pick non-trivial filters
        buf = io.BytesIO()
        Image.fromarray(big).save(buf, format="PNG", optimize=True)
        assert np.array_equal(decode_png_rgb8(buf.getvalue()), big)
        sc = load_from_gltf_native(_textured_gltf(tmp_path, "pillow", buf.getvalue(), "glb"))
        assert np.array_equal(sc.textures[sc.materials[sc.meshes[0].material].texture].image, big)
    # refused inputs
    lib = F.load_library()
    handle = C.c_void_p()
    deep = encode_png(rng.integers(0, 65536, (3, 3, 3)), 2, 16)
    assert lib.rt_gltf_load(os.fsencode(_textured_gltf(tmp_path, "deep", deep, "uri")), C.byref(handle)) == F.RT_ERR_PARSE
    assert b"16-bit" in lib.rt_last_error()
    with pytest.raises(ValueError):
        decode_png_rgb8(deep)
    jpeg = b"\xff\xd8\xff\xe0" + b"\0" * 32
    assert lib.rt_gltf_load(os.fsencode(_textured_gltf(tmp_path, "jpeg", jpeg, "file")), C.byref(handle)) == F.RT_ERR_PARSE
    assert b"JPEG" in lib.rt_last_error()
    broken = cases[0][1][:-40]
    assert lib.rt_gltf_load(os.fsencode(_textured_gltf(tmp_path, "broken", broken, "uri")), C.byref(handle)) == F.RT_ERR_PARSE


def test_jpeg_base_colour_textures(tmp_path):
    """JPEG base-colour textures (`gltf::import` -> `load_image`, src/scene.rs:87-114, 138-149).  The reference holds no
    JPEG fixture and its decoder (zune-jpeg 0.4.11) cannot run here, so the texels are parity-unpinned; what is
    checked: (1) the stages the decoder states — exact inverse transform of flat blocks, the integer YCbCr -> RGB
    formula, the 3:1 linear chroma upsampling with repeated edges — on files whose coefficients are known exactly;
    (2) agreement with libjpeg-turbo (through Pillow) within a few code values for baseline and progressive files,
    every chroma layout Pillow writes, restart intervals, odd sizes; (3) refusals; (4) the way through both glTF readers."""
    import io

    Image = pytest.importorskip("PIL.Image")
    from raytracer_amd.scene import decode_image_rgb8, load_from_gltf, load_from_gltf_native

    def jpeg(array, mode=None, **kw):
        buf = io.BytesIO()
        Image.fromarray(array, mode).save(buf, format="JPEG", **kw)
        return buf.getvalue()

    def clamp(x):
        return np.clip(x, 0, 255)

    def to_rgb(y, cb, cr):
        y, cb, cr = y.astype(np.int64), cb.astype(np.int64) - 128, cr.astype(np.int64) - 128
        return np.stack([clamp(y + ((45 * cr) >> 5)), clamp(y - ((11 * cb + 23 * cr) >> 5)), clamp(y + ((113 * cb) >> 6))], -1)

    # (1a) flat grey blocks at quality 100 (all quantisers 1) hold DC = 8 * (v - 128) exactly: every level comes back
    levels = np.arange(256, dtype=np.uint8).reshape(16, 16)
    flat = np.kron(levels, np.ones((8, 8), dtype=np.uint8))
    got = decode_image_rgb8(jpeg(flat, quality=100))
    assert got.shape == (128, 128, 3) and np.array_equal(got, np.repeat(flat[..., None], 3, 2))
    # (1b) the colour formula on flat YCbCr blocks, 4:4:4
    rng = np.random.default_rng(21)
    ycc = rng.integers(0, 256, (6, 9, 3)).astype(np.uint8)
    ycc_img = np.kron(ycc, np.ones((8, 8, 1), dtype=np.uint8))
    got = decode_image_rgb8(jpeg(ycc_img, "YCbCr", quality=100, subsampling=0))
    assert np.array_equal(got, to_rgb(ycc_img[..., 0], ycc_img[..., 1], ycc_img[..., 2]))
    # (1c) chroma upsampling: chroma constant over 16x16 (4:2:0), 16x8 (4:2:2) regions is coded exactly as one flat
    # 8x8 block per region; the decoder must interpolate that plane 3:1, vertically first, rounding after each step
    def up_rows(p):
        a = np.vstack([p[:1], p[:-1]]).astype(np.int64)
        b = np.vstack([p[1:], p[-1:]]).astype(np.int64)
        out = np.empty((2 * p.shape[0], p.shape[1]), dtype=np.int64)
        out[0::2] = (3 * p + a + 2) >> 2
        out[1::2] = (3 * p + b + 2) >> 2
        return out

    def up_cols(p):
        out = up_rows(p.T).T.copy()
        out[:, 0] = p[:, 0]
        out[:, -1] = p[:, -1]
        return out

    for sub, (fy, fx) in ((2, (2, 2)), (1, (1, 2))):
        coarse = rng.integers(0, 256, (4, 5, 3)).astype(np.uint8)
        img = np.kron(coarse, np.ones((8 * fy, 8 * fx, 1), dtype=np.uint8))
        got = decode_image_rgb8(jpeg(img, "YCbCr", quality=100, subsampling=sub))
        planes = []
        for c in (1, 2):
            p = np.kron(coarse[..., c], np.ones((8, 8), dtype=np.uint8)).astype(np.int64)
            if fy == 2:
                p = up_rows(p)
            planes.append(up_cols(p))
        assert np.array_equal(got, to_rgb(img[..., 0], planes[0], planes[1])), sub

    # (2) against libjpeg-turbo
    def picture(h, w):
        y, x = np.mgrid[0:h, 0:w]
        img = np.stack([127 + 120 * np.sin(x / 9.0 + y / 17.0), 127 + 120 * np.cos(x / 5.0 - y / 11.0), (x * 3 + y * 2) % 256], -1)
        return np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)

    worst = 0
    for (h, w) in ((64, 64), (37, 53), (1, 1), (17, 200)):
        img = picture(h, w)
        for sub in (0, 1, 2):
            for progressive in (False, True):
                for restart in (0, 3):
                    data = jpeg(img, quality=85, subsampling=sub, progressive=progressive, restart_marker_blocks=restart)
                    want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(np.int64)
                    got = decode_image_rgb8(data).astype(np.int64)
                    assert got.shape == want.shape
                    d = np.abs(got - want)
                    worst = max(worst, int(d.max()))
                    assert d.max() <= 5 and d.mean() <= 0.9, (h, w, sub, progressive, restart, d.max(), d.mean())
        data = jpeg(img[..., 0], quality=85)
        want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(np.int64)
        assert np.abs(decode_image_rgb8(data).astype(np.int64) - want).max() <= 1  # transform only
    assert worst >= 1  # the colour conversion is the approximate integer one, not libjpeg's

    # (3) refusals and damaged files: an error, never a crash
    with pytest.raises(ValueError, match="CMYK"):
        decode_image_rgb8(jpeg(np.zeros((8, 8, 4), dtype=np.uint8), "CMYK"))
    good = jpeg(picture(40, 40), quality=80, subsampling=2, progressive=True)
    with pytest.raises(ValueError, match="JPEG"):
        decode_image_rgb8(good[:20])
    with pytest.raises(ValueError, match="arithmetic"):
        decode_image_rgb8(good.replace(b"\xff\xc2", b"\xff\xca", 1))
    for cut in range(2, len(good), 37):
        try:
            decode_image_rgb8(good[:cut])
        except ValueError:
            pass

    # (4) through both glTF readers, in a GLB buffer view and as an external file
    img = picture(24, 40)
    data = jpeg(img, quality=90, subsampling=2)
    want = decode_image_rgb8(data)
    for how in ("glb", "file", "uri"):
        path = _textured_gltf(tmp_path, f"jpeg_{how}", data, how)
        for sc in (load_from_gltf_native(path), load_from_gltf(path)):
            t = sc.textures[sc.materials[sc.meshes[0].material].texture]
            assert t.kind == F.RT_TEX_IMAGE and np.array_equal(t.image, want)


def test_rust_binding_declares_every_entry_point():
    """The `-sys` crate source (bindings/rust, uncompiled: no Rust toolchain here) must stay in step with the
    header: same set of functions, same field order in every struct."""
    header = open(os.path.join(ROOT, "include", "rt_mi355x.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    rust = open(os.path.join(ROOT, "bindings", "rust", "rt-mi355x-sys", "src", "lib.rs")).read()
    declared = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", header))
    bound = set(re.findall(r"pub fn (rt_[a-z_0-9]+)\s*\(", rust))
    assert declared == bound, declared ^ bound
    for name, body in re.findall(r"typedef struct (rt_[a-z_]+) \{(.*?)\} \1;", header, flags=re.S):
        c_fields = re.findall(r"([a-z_0-9]+)(?:\[\d+\])?;", body)
        m = re.search(r"pub struct %s \{(.*?)\n\}" % name, rust, flags=re.S)
        assert m, f"{name} missing from the Rust binding"
        rust_fields = re.findall(r"pub ([a-z_0-9]+):", m.group(1))
        assert c_fields == rust_fields, (name, c_fields, rust_fields)


def test_benchmark_scenes_are_pinned():
    """The procedural benchmark scenes (the 485 spheres, the triangle soups, the lit teapot) and the two glTF assets hash
    to the digests committed in tests/golden/scene_digests.json (made by tests/golden/make_scene_digests.py): the
    workloads behind the bench legs cannot drift with the installed numpy or an edit of the generators.  The soup draws
    from SplitMix64 twelve values per triangle, so its first million triangles do not depend on the scene's size."""
    import json

    from raytracer_amd.scenes import SplitMix64, splitmix64_units

    sys_path = os.path.join(ROOT, "tests", "golden")
    want = json.load(open(os.path.join(sys_path, "scene_digests.json")))
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_scene_digests", os.path.join(sys_path, "make_scene_digests.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.digests() == want
    # the array-wise generator is the sequential one
    rng = SplitMix64(0xC0FFEE)
    seq = np.array([rng.unit() for _ in range(300)], np.float32)
    assert np.array_equal(splitmix64_units(0xC0FFEE, 0, 300), seq)
    assert np.array_equal(splitmix64_units(0xC0FFEE, 123, 100), seq[123:223])
