"""Shared test configuration: fixture scenes, benchmark poses, helpers."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODELS = os.path.join(ROOT, "models")

# name -> (file, camera position, camera forward, path depth used by BASELINE.json's configs)
SCENES = {
    "cornell_box": ("cornell_box.glb", (3.6, 1.25, 0.0), (-1.0, 0.0, 0.0), 8),
    "suzanne_plane": ("suzanne_plane.glb", (0.0, 1.2, 4.5), (0.0, -0.15, -1.0), 2),
    "dense_monkey": ("dense_monkey.glb", (0.0, 0.0, 2.7), (0.0, 0.0, -1.0), 2),
    "room": ("room.glb", (0.0, 1.4, 2.0), (0.0, 0.0, -1.0), 16),
    "uv_texture_test": (os.path.join("test", "UVtextureTest.glb"), (0.0, 1.0, 4.0), (0.0, -0.1, -1.0), 3),
    "bvh_split_test": (os.path.join("test", "bvhsplitTest.glb"), (0.0, 2.0, 5.0), (0.0, -0.2, -1.0), 2),
    "multi_material": (os.path.join("test", "multiMaterialMeshTest.glb"), (0.0, 2.0, 5.0), (0.0, -0.2, -1.0), 2),
    # 16 textured materials, 6 RGBA (palettised PNG + tRNS -> 4 channels): the alpha cut-out scene of the reference's only
    # published timings (beforeBVHbuildrefactor_col.png: 843x460, 5 bounces)
    "mc_transparency": (os.path.join("test", "mcTransparencyTest.glb"), (0.0, 3.0, 9.0), (0.0, -0.15, -1.0), 5),
    # a full Counter-Strike map from the reference's models/source (11 167 triangles, 23 textured materials, closed
    # geometry: every path runs to the bounce limit): the large-scene case whose BVH and triangles do not fit in LDS
    "cs16_dust": ("cs16_dust.glb", (-11.4, 1.5, -3.85), (0.0, 0.0, 1.0), 5),
    # the reference's sun-shadow test: a plane with a 2164x2152 baseline-JPEG texture under a cube (models/test)
    "sunshadow_test": (os.path.join("test", "sunshadowTest.glb"), (0.0, 3.0, 7.0), (0.0, -0.35, -1.0), 3),
    # the ASCII glTF branch (Scene.cu:38-41): external .bin, texture file named by uri.  cornell_box.gltf is a 36-triangle
    # variant of the box with a ceiling light mesh and a different texture; suzanne_plane.gltf has node TRS (ignored)
    "cornell_box_gltf": ("cornell_box.gltf", (3.6, 1.25, 0.0), (-1.0, 0.0, 0.0), 4),
    "uv_texture_gltf": ("UVtextureTest.gltf", (0.0, 1.0, 4.0), (0.0, -0.1, -1.0), 3),
    "suzanne_plane_gltf": ("suzanne_plane.gltf", (0.0, 1.2, 4.5), (0.0, -0.15, -1.0), 2),
    "lightweight_rt": (os.path.join("test", "lightweightRTtest.glb"), (0.0, 1.8, 7.5), (0.0, -0.1, -1.0), 3),
    # the reference's own scene-hierarchy test scene (models/sceneHierTest.glb): 12 nodes / meshes, index accessors shared by
    # several meshes, three embedded images
    # the reference's test scene for the material features it loads and never reads: three emissive cubes (blue, red,
    # green), a metallic sphere of roughness 0, a teapot on a textured floor
    "emissive_test": (os.path.join("test", "EmissiveTest.glb"), (0.0, 0.9, 2.2), (0.0, -0.3, -1.0), 4),
    "scene_hier_test": ("sceneHierTest.glb", (6.0, 3.0, 7.0), (-0.6, -0.2, -0.7), 3),
}


def scene_path(name):
    return os.path.join(MODELS, SCENES[name][0])


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)
