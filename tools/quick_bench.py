"""Ad-hoc GPU timing of one scene: python tools/quick_bench.py [cornell|teapot] W H SPP [flags]."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from raytracer_amd import RenderSettings, _ffi as F, load_from_gltf
from raytracer_amd.renderer import Context, Renderer, Scene, camera_from_settings


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
    h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
    spp = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    pipeline = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    max_paths = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    from raytracer_amd.scenes import lit_teapot, rtiow_final_scene, triangle_soup
    bg = (0.0, 0.0, 0.0)
    if name == "teapot_lit":
        desc, cs = lit_teapot(load_from_gltf(os.path.join(ROOT, "assets", "teapot.gltf")))
    elif name == "rtiow":
        desc, cs = rtiow_final_scene()
        bg = (0.7, 0.8, 1.0)
    elif name.startswith("soup"):
        desc, cs = triangle_soup(int(name[4:]))
    else:
        desc = load_from_gltf(os.path.join(ROOT, "assets", f"{name}.gltf"))
        cs = desc.camera(0)
    ctx = Context(0, flags=flags)
    ctx.set_pipeline(pipeline)
    ctx.set_max_paths(max_paths)
    t0 = time.time()
    scene = Scene(ctx, desc)
    i = scene.info()
    print(f"scene {name}: {i.num_triangles} tris, {i.num_nodes} nodes, depth {i.depth}, build+upload {time.time()-t0:.3f}s")
    cam = camera_from_settings(cs, w, h)
    rs = RenderSettings(w, h, 0, 50, spp, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=1, background_color=bg)
    r = Renderer(cam, scene, rs)
    for rep in range(3):
        t0 = time.time()
        img = r.render_sums()
        dt = time.time() - t0
        st = r.last_stats
        print(f"rep {rep}: wall {dt*1e3:.1f} ms, gpu {st.total_ms:.1f} ms, rays {st.rays_traced} paths {st.paths} "
              f"rays/path {st.rays_traced/st.paths:.3f} -> {st.rays_traced/st.total_ms/1e3:.1f} Mrays/s; "
              f"pipe {st.pipeline} bounce {st.bounce_ms:.2f} ({st.timed_bounce_rays/1e6:.1f}M) ext_rays {st.timed_extend_rays/1e6:.1f}M gen {st.generate_ms:.2f} ext {st.extend_ms:.2f} shade {st.shade_ms:.2f} acc {st.accumulate_ms:.2f} ms; "
              f"nodes/ray {st.node_visits/max(1,st.rays_traced):.2f} tri/ray {st.triangle_tests/max(1,st.rays_traced):.2f}")
    if hasattr(ctx.lib, "rt_debug_lane_stats"):
        out = (C.c_ulonglong * 16)()
        ctx.lib.rt_debug_lane_stats(out, 1)
        v = list(out)
        rounds = v[0] + v[2]
        print(f"lane stats: node rounds {v[0]} ({v[1]/max(1,v[0]):.1f} lanes), leaf rounds {v[2]} ({v[3]/max(1,v[2]):.1f} lanes); "
              f"per round: finished {v[4]/max(1,rounds):.1f}, without path {v[5]/max(1,rounds):.1f}, waiting at leaf in node rounds "
              f"{v[6]/max(1,v[0]):.1f}; services {v[7]} ({v[8]/max(1,v[7]):.1f} lanes shaded)")
    img /= spp
    fin = np.isfinite(img)
    print(f"nan {np.isnan(img).mean():.4f} inf {np.isinf(img).mean():.4f} mean(finite) {img[fin].mean():.5f}")


if __name__ == "__main__":
    main()
