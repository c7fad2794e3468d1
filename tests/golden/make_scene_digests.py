"""Writes tests/golden/scene_digests.json: the sha256 digests (raytracer_amd.scenes.scene_digest) that pin the
procedural benchmark scenes.  Run from the repository root: python tests/golden/make_scene_digests.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from raytracer_amd import load_from_gltf  # noqa: E402
from raytracer_amd.scenes import lit_teapot, rtiow_final_scene, scene_digest, triangle_soup  # noqa: E402


def digests():
    out = {}
    out["rtiow_final_scene"] = scene_digest(rtiow_final_scene()[0])
    out["triangle_soup_100000"] = scene_digest(triangle_soup(100_000)[0])
    soup = triangle_soup(10_000_000)[0]
    out["triangle_soup_10000000_first_1000000_triangles"] = scene_digest(soup, 1_000_000)
    out["triangle_soup_10000000"] = scene_digest(soup)
    del soup
    out["cornell_gltf"] = scene_digest(load_from_gltf(os.path.join(ROOT, "assets", "cornell.gltf")))
    teapot = load_from_gltf(os.path.join(ROOT, "assets", "teapot.gltf"))
    out["teapot_gltf"] = scene_digest(teapot)
    out["lit_teapot"] = scene_digest(lit_teapot(teapot)[0])
    return out


if __name__ == "__main__":
    path = os.path.join(ROOT, "tests", "golden", "scene_digests.json")
    json.dump(digests(), open(path, "w"), indent=1)
    print(open(path).read())
