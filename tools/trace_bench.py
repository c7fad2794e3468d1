"""Ad-hoc GPU timing of rt_trace_closest: python tools/trace_bench.py [cornell|teapot|soupN] [million rays]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from raytracer_amd import _ffi as F, load_from_gltf
from raytracer_amd.renderer import Context, Scene


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
    n = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 8_000_000
    from raytracer_amd.scenes import triangle_soup
    if name.startswith("soup"):
        desc, _ = triangle_soup(int(name[4:]))
        lo, hi = np.full(3, -1.0), np.full(3, 1.0)
    else:
        desc = load_from_gltf(os.path.join(ROOT, "assets", f"{name}.gltf"))
        v = np.concatenate([np.asarray(m.positions).reshape(-1, 3) for m in desc.meshes])
        lo, hi = v.min(0), v.max(0)
    ctx = Context(0, flags=F.RT_CTX_TIMING)
    scene = Scene(ctx, desc)
    rng = np.random.default_rng(7)
    rays = np.zeros(n, dtype=F.RAY_DTYPE)
    o = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["origin"] = o
    rays["direction"] = d
    for rep in range(3):
        hits, st = scene.trace_closest(rays, 1e-4, float("inf"))
        print(f"rep {rep}: extend {st.extend_ms:.2f} ms -> {n / st.extend_ms / 1e3:.1f} Mrays/s; hit fraction "
              f"{(hits['prim_id'] != 0xFFFFFFFF).mean():.3f}")


if __name__ == "__main__":
    main()
