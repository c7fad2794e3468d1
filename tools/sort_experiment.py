"""Measures how much ray ordering matters for closest-hit queries on a large scene: the same secondary rays
traced in pixel order, shuffled, and sorted by a Morton key of the origin (optionally with the direction
octant).  usage: python tools/sort_experiment.py [n_triangles] [W] [H]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

from raytracer_amd import _ffi as F
from raytracer_amd.renderer import Context, Scene, camera_from_settings
from raytracer_amd.scenes import triangle_soup
import helpers


def morton3(q):
    def spread(v):
        v = v.astype(np.uint64) & 0x3FF
        v = (v | (v << 16)) & 0x30000FF
        v = (v | (v << 8)) & 0x300F00F
        v = (v | (v << 4)) & 0x30C30C3
        v = (v | (v << 2)) & 0x9249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def main():
    n_tri = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
    h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
    desc, cs = triangle_soup(n_tri)
    ctx = Context(0)
    scene = Scene(ctx, desc)
    lib = scene.lib
    cam = camera_from_settings(cs, w, h)
    pr = helpers.primary_rays(cam, w, h, 1)
    hits, _ = scene.trace_closest(pr, 0.1, 1000.0)
    ok = hits["prim_id"] != F.RT_MISS
    p = (pr["origin"][ok] + pr["direction"][ok] * hits["t"][ok][:, None]).astype(np.float32)
    rng = np.random.default_rng(1)
    d = rng.normal(size=p.shape).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros(len(p), dtype=F.RAY_DTYPE)
    rays["origin"] = p
    rays["direction"] = d
    print(f"{len(rays)} secondary rays")
    q = np.clip((p + 1.0) * 0.5 * 1023.0, 0, 1023).astype(np.uint32)
    key = morton3(q)
    octant = ((d[:, 0] < 0).astype(np.uint64) | ((d[:, 1] < 0).astype(np.uint64) << 1) | ((d[:, 2] < 0).astype(np.uint64) << 2))
    orders = {
        "pixel order": np.arange(len(rays)),
        "shuffled": rng.permutation(len(rays)),
        "morton(origin)": np.argsort(key, kind="stable"),
        "octant|morton": np.argsort((octant << 30) | key, kind="stable"),
        "morton|octant": np.argsort((key << 3) | octant, kind="stable"),
        "morton15|octant": np.argsort(((key >> 15) << 3) | octant, kind="stable"),
    }
    out = torch.empty(len(rays) * 32, dtype=torch.uint8, device="cuda")
    ref = None
    for name, order in orders.items():
        r = torch.from_numpy(rays[order].view(np.uint8).reshape(-1)).cuda()
        ms = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            F.check(lib, lib.rt_trace_closest_device(ctx.handle, scene.handle, r.data_ptr(), len(rays), 0.1, 1000.0, 1,
                                                     out.data_ptr(), None))
            ctx.synchronize()
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        res = out.cpu().numpy().view(F.HIT_DTYPE)
        inv = np.empty_like(order)
        inv[order] = np.arange(len(order))
        res = res[inv]
        if ref is None:
            ref = res.copy()
        same = bool((res["prim_id"] == ref["prim_id"]).all() and (res["t"].view(np.uint32) == ref["t"].view(np.uint32)).all())
        print(f"{name:18s}: {min(ms):8.2f} ms  -> {len(rays)/min(ms)/1e3:8.1f} Mrays/s   identical={same}")


if __name__ == "__main__":
    main()
