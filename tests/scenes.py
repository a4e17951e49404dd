"""Scene definitions shared by the tests (data of the reference's shipped scenes + small synthetic ones).

Scene preparation here goes through the ORACLE's restatement of the reference's cold path
(Camera::new, mesh SoA conversion) — fine for tests; product code paths use the C++ host instead.
"""
from __future__ import annotations

import numpy as np

from rbrt_amd import abi, standin

# scenes/example_scene.yaml:2-15 (same camera in scenes/header_card.yaml)
CAMERA = dict(position=(0.0, 5.0, 4.0), look_at=(0.0, -0.1, -1.0), up=(0.0, 1.0, -0.4), focal_mm=28.0)

L, M, D = abi.MAT_LAMBERTIAN, abi.MAT_METAL, abi.MAT_DIELECTRIC

# scenes/example_scene.yaml:33-75
EXAMPLE_SPHERES = [
    ((0.0, -1000.0, -5.0), 1000.0, abi.material(L, (0.02, 0.2, 0.1))),
    ((-5.0, 1.5, -9.0), 1.5, abi.material(L, (0.1, 0.1, 0.9))),
    ((-2.5, 2.9, -15.0), 3.0, abi.material(M, (0.8, 0.8, 0.8), 0.005)),
    ((1.5, 1.25, -9.0), 1.5, abi.material(D, (0, 0, 0), 1.8)),
]
# scenes/example_scene.yaml:17-31
EXAMPLE_MESH = dict(scale=45.0, translation=(5.0, -1.8, -12.5), rotation=(0.0, 0.0, 0.0),
                    mat=abi.material(D, (0.8, 0.8, 0.8), 0.2))

# scenes/header_card.yaml:33-105
HEADER_SPHERES = [
    ((0.0, -1000.0, -12.0), 1000.0, abi.material(L, (0.02, 0.2, 0.1))),
    ((-7.5, 1.5, -10.5), 1.5, abi.material(L, (0.1, 0.1, 0.9))),
    ((3.0, 0.7, -9.5), 0.7, abi.material(L, (0.5, 0.5, 0.1))),
    ((-4.5, 2.5, -16.0), 2.5, abi.material(M, (0.9, 0.9, 0.9), 0.005)),
    ((9.5, 4.0, -20.0), 4.0, abi.material(M, (0.9, 0.9, 0.9), 0.001)),
    ((-1.5, 1.0, -8.0), 1.0, abi.material(D, (0, 0, 0), 1.8)),
    ((7.0, 1.5, -10.0), 1.5, abi.material(D, (0, 0, 0), 1.8)),
]
# scenes/header_card.yaml:17-31
HEADER_MESH = dict(scale=45.0, translation=(3.5, -1.8, -14.0), rotation=(0.0, 0.0, 0.0),
                   mat=abi.material(L, (1.0, 0.0, 0.0)))


def camera(oracle, width, height, **over):
    c = dict(CAMERA)
    c.update(over)
    return oracle.camera_new(c["position"], c["look_at"], c["up"], height, width, c["focal_mm"])


def spheres_scene(spheres=EXAMPLE_SPHERES):
    return abi.SceneData(spheres=spheres)


def standin_mesh(oracle, n_triangles, scale, translation, rotation, mat, kind="smooth"):
    return oracle.mesh_prep(standin.triangles(n_triangles, kind), scale, rotation, translation, mat)


def example_scene(oracle, n_triangles=standin.BUNNY_TRIANGLES, mesh_over=None, spheres=EXAMPLE_SPHERES, kind="smooth"):
    """scenes/example_scene.yaml with the stand-in mesh (n_triangles of it; kind: the smooth blob or the rough one)."""
    m = dict(EXAMPLE_MESH)
    m.update(mesh_over or {})
    return abi.SceneData(spheres=spheres, meshes=[standin_mesh(oracle, n_triangles, kind=kind, **m)])


def header_scene(oracle, n_triangles=standin.BUNNY_TRIANGLES):
    return abi.SceneData(spheres=HEADER_SPHERES, meshes=[standin_mesh(oracle, n_triangles, **HEADER_MESH)])


def random_soup(rng: np.random.Generator, n, extent=1.0, size=0.3):
    """n random triangles (n,3,3) float32 inside a cube of half-size `extent`."""
    c = rng.uniform(-extent, extent, (n, 1, 3))
    return (c + rng.uniform(-size, size, (n, 3, 3))).astype(np.float32)


# A scene with BasicTriangle elements (triangle.rs:9-34; the reference's YAML cannot express them, its API can):
# the example scene's spheres + a metal mirror quad (two triangles) behind them, a glass triangle in front of the camera
# and a diffuse one lying on the ground. Elements are INTERLEAVED with the spheres (Scene::elements order matters for
# ties only, but object ids follow it).
TRIANGLES = [
    (((-9.0, 0.0, -19.0), (3.0, 0.0, -21.0), (3.0, 7.0, -21.0)), abi.material(M, (0.9, 0.85, 0.8), 0.02)),
    (((-9.0, 0.0, -19.0), (3.0, 7.0, -21.0), (-9.0, 7.0, -19.0)), abi.material(M, (0.9, 0.85, 0.8), 0.02)),
    (((-1.5, 0.3, -5.5), (1.0, 0.4, -6.0), (-0.2, 2.6, -6.5)), abi.material(D, (0, 0, 0), 1.5)),
    (((2.0, 0.02, -6.0), (5.5, 0.02, -7.0), (3.0, 0.02, -9.5)), abi.material(L, (0.8, 0.3, 0.1))),
]
T = 0x80000000
TRIANGLE_ORDER = [0, T | 0, 1, T | 2, 2, T | 1, 3, T | 3]  # sphere 0, tri 0, sphere 1, tri 2, sphere 2, tri 1, sphere 3, tri 3


def triangle_scene(oracle=None, n_mesh_triangles=0, order=TRIANGLE_ORDER):
    meshes = [standin_mesh(oracle, n_mesh_triangles, **EXAMPLE_MESH)] if n_mesh_triangles else []
    return abi.SceneData(spheres=EXAMPLE_SPHERES, meshes=meshes, triangles=TRIANGLES, element_order=order)
