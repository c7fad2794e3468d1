"""MI355X-native path-tracing core behind the GyrosOfWar/raytracer Renderer surface."""
from . import _ffi  # noqa: F401
from .scene import (CameraSettings, Material, RenderSettings, SceneDescription, Sphere, Texture,  # noqa: F401
                    TriangleMesh, load_from_gltf)
