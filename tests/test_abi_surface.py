"""The C-ABI library loads without a GPU and exports every symbol include/drt.h declares; the reference-shaped C++
wrapper (include/DustRayTracer.hpp) compiles and links against it.  No compute calls here."""
import ctypes
import os
import re
import subprocess

import pytest

from tests.scenes import ROOT

drt = pytest.importorskip("dustraytracer_amd")


def declared_functions():
    text = open(os.path.join(ROOT, "include", "drt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(drt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    names = declared_functions()
    assert len(names) >= 45
    lib = ctypes.CDLL(drt.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_covers_the_header():
    src = open(os.path.join(ROOT, "dustraytracer_amd", "__init__.py")).read()
    unbound = [n for n in declared_functions() if n not in src and not re.search(r'"drt_scene_%s_count"', n)]
    unbound = [n for n in unbound if not re.match(r"drt_scene_(triangle|node|material|texture|mesh)_count|drt_scene_get_(triangles|nodes|materials|meshes)|drt_renderer_(width|height|sample_count|local_rows)", n)]
    assert not unbound, unbound


def test_abi_version_and_defaults_without_gpu():
    assert drt._lib.drt_abi_version() == 2
    s = drt.RendererSettings()
    assert (s.max_samples, s.ray_bounce_limit, s.gamma_correction, s.tone_mapping, s.enableSunlight) == (500, 2, 1, 1, 0)
    assert abs(s.sky_intensity - 20) < 1e-6 and abs(s.sunlight_intensity - 30) < 1e-6
    cam = drt.Camera()
    assert abs(cam.vfov_rad - 1.0471975) < 1e-6 and cam.focus_dist == 10 and list(cam.m_Position) == [0, 2, 5]
    assert drt.shard_rows(1080, 8, 0, 8) == 136 and drt.shard_rows(1080, 8, 7, 8) == 128


def test_no_gpu_means_an_error_not_a_fallback():
    if drt.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(drt.DrtError) as e:
        drt.Renderer(0)
    assert e.value.code == drt.ERR_DEVICE


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "dustraytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                for line in open(os.path.join(dirpath, f), errors="replace"):
                    code = line.split("//")[0].split("#" if f.endswith(".py") else "\0")[0] if not line.lstrip().startswith("#include") else line
                    assert not re.search(r"^\s*(import|from)\s+oracle\b", code), (f, line)
                    assert not (line.lstrip().startswith("#include") and "oracle" in line), (f, line)
                    assert "dlopen" not in code or "oracle" not in code, (f, line)


def test_cpp_wrapper_compiles_and_links(tmp_path):
    exe = tmp_path / "drt_render"
    cmd = ["g++", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "drt_render.cpp"),
           "-L" + os.path.join(ROOT, "dustraytracer_amd"), "-ldrt_hip", "-Wl,-rpath," + os.path.join(ROOT, "dustraytracer_amd"),
           "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)          # no arguments: usage, exit code 2
    assert r.returncode == 2 and "usage" in r.stderr


def _ref_layout():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "ref_layout.json")))


def test_pod_layouts_match_the_reference_headers(tmp_path):
    """include/drt.h's PODs have the sizes and field offsets of the reference's own structs (Vertex, Triangle, BVHNode,
    Material as its compiler lays them out: tests/golden/ref_layout.json, written by oracle/_ref/ref_kat `layout`), so
    m_PrimitivesBuffer / m_BVHNodes / m_Material can be handed over as they are."""
    src = tmp_path / "layout.c"
    src.write_text(r"""
#include <stddef.h>
#include <stdio.h>
#include "drt.h"
#define P(k, v) printf("%s=%lld\n", k, (long long)(v))
int main(void) {
    P("sizeof_Vertex", sizeof(drt_vertex)); P("Vertex.position", offsetof(drt_vertex, position)); P("Vertex.normal", offsetof(drt_vertex, normal)); P("Vertex.UV", offsetof(drt_vertex, uv));
    P("sizeof_Triangle", sizeof(drt_triangle)); P("Triangle.centroid", offsetof(drt_triangle, centroid)); P("Triangle.vertex0", offsetof(drt_triangle, vertex[0]));
    P("Triangle.vertex1", offsetof(drt_triangle, vertex[1])); P("Triangle.vertex2", offsetof(drt_triangle, vertex[2]));
    P("Triangle.face_normal", offsetof(drt_triangle, face_normal)); P("Triangle.materialIdx", offsetof(drt_triangle, material));
    P("sizeof_BVHNode", sizeof(drt_bvh_node)); P("BVHNode.m_IsLeaf", offsetof(drt_bvh_node, is_leaf)); P("BVHNode.m_BoundingBox", offsetof(drt_bvh_node, bmin));
    P("BVHNode.dev_child1_idx", offsetof(drt_bvh_node, child1)); P("BVHNode.dev_child2_idx", offsetof(drt_bvh_node, child2));
    P("BVHNode.primitives_count", offsetof(drt_bvh_node, prim_count)); P("BVHNode.primitive_start_idx", offsetof(drt_bvh_node, prim_start));
    P("sizeof_Material", sizeof(drt_material)); P("Material.Albedo", offsetof(drt_material, albedo)); P("Material.EmmisiveFactor", offsetof(drt_material, emissive));
    P("Material.AlbedoTextureIndex", offsetof(drt_material, albedo_tex)); P("Material.Roughness", offsetof(drt_material, roughness));
    P("Material.Transmission", offsetof(drt_material, transmission)); P("Material.refractive_index", offsetof(drt_material, refractive_index));
    P("Material.Metallic", offsetof(drt_material, metallic));
    return 0;
}
""")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    ours = dict(line.split("=") for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split())
    ref = _ref_layout()
    assert len(ours) == 26
    for k, v in ours.items():
        assert int(v) == ref[k], (k, v, ref[k])
    # the numpy views the Python mirror hands out use the same offsets
    assert drt.TRIANGLE_DTYPE.itemsize == ref["sizeof_Triangle"] and drt.TRIANGLE_DTYPE.fields["vertex"][1] == ref["Triangle.vertex0"]
    assert drt.TRIANGLE_DTYPE.fields["face_normal"][1] == ref["Triangle.face_normal"] and drt.TRIANGLE_DTYPE.fields["material"][1] == ref["Triangle.materialIdx"]
    assert drt.NODE_DTYPE.itemsize == ref["sizeof_BVHNode"] and drt.NODE_DTYPE.fields["child1"][1] == ref["BVHNode.dev_child1_idx"]
    assert drt.NODE_DTYPE.fields["prim_start"][1] == ref["BVHNode.primitive_start_idx"]
    assert drt.MATERIAL_DTYPE.itemsize == ref["sizeof_Material"] and drt.MATERIAL_DTYPE.fields["albedo_tex"][1] == ref["Material.AlbedoTextureIndex"]


def test_defaults_match_the_reference_headers():
    """drt_default_settings / drt_default_camera / Camera() = the default member initialisers of RendererSettings.h:22-34
    and Camera.cuh:32-46, compared as bit patterns."""
    import numpy as np
    ref = _ref_layout()
    f = lambda x: int(np.float32(x).view(np.uint32))
    s = drt.RendererSettings()
    for k, v in [("gamma_correction", s.gamma_correction), ("tone_mapping", s.tone_mapping), ("enableSunlight", s.enableSunlight),
                 ("max_samples", s.max_samples), ("ray_bounce_limit", s.ray_bounce_limit), ("RenderMode", s.RenderMode), ("DebugMode", s.DebugMode)]:
        assert int(v) == ref["settings." + k], k
    for k, v in [("sunlight_dir.x", s.sunlight_dir[0]), ("sunlight_dir.y", s.sunlight_dir[1]), ("sunlight_color.x", s.sunlight_color[0]),
                 ("sunlight_color.y", s.sunlight_color[1]), ("sunlight_color.z", s.sunlight_color[2]), ("sunlight_intensity", s.sunlight_intensity),
                 ("sky_color.x", s.sky_color[0]), ("sky_color.y", s.sky_color[1]), ("sky_color.z", s.sky_color[2]), ("sky_intensity", s.sky_intensity)]:
        assert f(v) == ref["settings." + k], k
    cam = drt.Camera()
    for k, v in [("exposure", cam.exposure), ("vfov_rad", cam.vfov_rad), ("defocus_angle", cam.defocus_angle), ("focus_dist", cam.focus_dist),
                 ("m_movement_speed", cam.m_movement_speed)]:
        assert f(v) == ref["camera." + k], k
    for name in ("m_Position", "m_Forward_dir", "m_Up_dir", "m_Right_dir"):
        for i, axis in enumerate("xyz"):
            assert f(getattr(cam, name)[i]) == ref["camera.%s.%s" % (name, axis)], (name, axis)


def test_cpp_wrapper_defaults_match_the_reference_headers(tmp_path):
    """include/DustRayTracer.hpp: RendererSettings{} and Camera{} carry the reference's default member values."""
    src = tmp_path / "defaults.cpp"
    src.write_text(r"""
#include <cstdio>
#include <cstring>
#include "DustRayTracer.hpp"
static void kv(const char *k, long long v) { std::printf("%s=%lld\n", k, v); }
static void kf(const char *k, float v) { unsigned u; std::memcpy(&u, &v, 4); kv(k, (long long)u); }
int main() {
    RendererSettings rs;
    kv("settings.gamma_correction", rs.gamma_correction); kv("settings.tone_mapping", rs.tone_mapping); kv("settings.enableSunlight", rs.enableSunlight);
    kv("settings.max_samples", rs.max_samples); kv("settings.ray_bounce_limit", rs.ray_bounce_limit);
    kv("settings.RenderMode", (int)rs.RenderMode); kv("settings.DebugMode", (int)rs.DebugMode);
    kf("settings.sunlight_dir.x", rs.sunlight_dir.x); kf("settings.sunlight_dir.y", rs.sunlight_dir.y);
    kf("settings.sunlight_color.x", rs.sunlight_color.x); kf("settings.sunlight_color.y", rs.sunlight_color.y); kf("settings.sunlight_color.z", rs.sunlight_color.z);
    kf("settings.sunlight_intensity", rs.sunlight_intensity);
    kf("settings.sky_color.x", rs.sky_color.x); kf("settings.sky_color.y", rs.sky_color.y); kf("settings.sky_color.z", rs.sky_color.z);
    kf("settings.sky_intensity", rs.sky_intensity);
    Camera cam;
    kf("camera.exposure", cam.exposure); kf("camera.vfov_rad", cam.vfov_rad); kf("camera.defocus_angle", cam.defocus_angle);
    kf("camera.focus_dist", cam.focus_dist); kf("camera.m_movement_speed", cam.m_movement_speed);
    kf("camera.m_Position.x", cam.m_Position.x); kf("camera.m_Position.y", cam.m_Position.y); kf("camera.m_Position.z", cam.m_Position.z);
    kf("camera.m_Forward_dir.x", cam.m_Forward_dir.x); kf("camera.m_Forward_dir.y", cam.m_Forward_dir.y); kf("camera.m_Forward_dir.z", cam.m_Forward_dir.z);
    kf("camera.m_Up_dir.x", cam.m_Up_dir.x); kf("camera.m_Up_dir.y", cam.m_Up_dir.y); kf("camera.m_Up_dir.z", cam.m_Up_dir.z);
    kf("camera.m_Right_dir.x", cam.m_Right_dir.x); kf("camera.m_Right_dir.y", cam.m_Right_dir.y); kf("camera.m_Right_dir.z", cam.m_Right_dir.z);
    kf("deg2rad_60", deg2rad(60));
    return 0;
}
""")
    exe = tmp_path / "defaults"
    lib_dir = os.path.dirname(drt.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), str(src), "-L" + lib_dir, "-ldrt_hip",
                    "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    ours = dict(line.split("=") for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split())
    ref = _ref_layout()
    assert len(ours) == 35
    for k, v in ours.items():
        assert int(v) == ref[k], (k, v, ref[k])


def _build_editor_calls(tmp_path):
    exe = tmp_path / "editor_calls"
    lib_dir = os.path.dirname(drt.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "editor_calls.cpp"),
                    "-L" + lib_dir, "-ldrt_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    return exe


def test_editor_statements_compile_against_the_wrapper(tmp_path):
    """tests/cpp/editor_calls.cpp repeats the statements Editor/EditorLayer.cpp makes on Scene / BVHBuilder / Camera /
    RendererSettings / Renderer (member names, argument types, the buildIterative(m_PrimitivesBuffer, m_BVHNodes) spelling,
    addresses of settings members for the widgets): it must compile warning-free, and its scene / camera part must run."""
    exe = _build_editor_calls(tmp_path)
    out = subprocess.run([str(exe), os.path.join(ROOT, "models", "cs16_dust.glb")], capture_output=True, text=True, check=True).stdout
    assert "objects=1 triangles=11167 materials=23 textures=23 root=1" in out     # the editor's metrics panel
    assert "settings: 0 1 1 2 500 0.944 30.0 -0.803 0.681 0.800 20.0 | 1.0472 10.0 0.0 1.0" in out
    # Core/Sampler.cuh's interface, implemented (PCGSampler): RayGen's seed recipe, and the reference's own randomFloat sequence
    assert "sampler: seed=285 pixel=57,0" in out
    assert "sampler: draws=0x1.e8bd5ap-1,0x1.55b33p-2,0x1.4d657ep-4" in out              # kat_ref.npz randfloat(12345)[:3]


@pytest.mark.gpu
def test_editor_statements_render(tmp_path):
    exe = _build_editor_calls(tmp_path)
    out = subprocess.run([str(exe), os.path.join(ROOT, "models", "cornell_box.glb"), "--render"], capture_output=True, text=True, check=True).stdout
    assert "rendered 160 x 90, sample 2" in out


def _build_gl_shim(tmp_path):
    exe = tmp_path / "editor_gl_shim"
    lib_dir = os.path.dirname(drt.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "editor_gl_shim.cpp"),
                    "-L" + lib_dir, "-ldrt_hip", "-ldl", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    return exe


def test_editor_gl_shim_displays_and_saves_like_the_editor(tmp_path):
    """include/DustRayTracerGL.hpp: the GL half of the render target the reference fills through CUDA-GL interop (Renderer.cu:42-94),
    the viewport texture (EditorLayer.cpp:293-295) and "save png" (EditorLayer.cpp:23-31, 85-96: glGetTexImage as RGBA8, rows
    flipped).  Compiled warning-free and run with a software texture in place of OpenGL: the PNG must hold the frame clamped to
    [0, 1], scaled to 8 bits and upside down."""
    import numpy as np
    from PIL import Image
    exe = _build_gl_shim(tmp_path)
    base = tmp_path / "image"
    out = subprocess.run([str(exe), str(base)], capture_output=True, text=True, check=True).stdout
    assert "shown texture 1, 37 x 21, uploads 2" in out and "Image saved" in out
    png = np.asarray(Image.open(str(base) + ".png"))
    assert png.shape == (21, 37, 4)
    W, H = 37, 21
    x = (np.arange(W, dtype=np.float32) / np.float32(W - 1) * np.float32(1.5) - np.float32(0.25))
    y = np.arange(H, dtype=np.float32) / np.float32(H - 1)
    want = np.zeros((H, W, 4), np.float32)
    want[..., 0], want[..., 1], want[..., 2], want[..., 3] = x[None, :], y[:, None], 0.5, 1.0
    want8 = np.rint(np.clip(want, 0, 1) * np.float32(255)).astype(np.uint8)[::-1]          # flipped: the frame's row 0 is the bottom
    assert np.array_equal(png, want8)


@pytest.mark.gpu
def test_editor_gl_shim_shows_a_rendered_frame(tmp_path):
    import numpy as np
    from PIL import Image
    exe = _build_gl_shim(tmp_path)
    base = tmp_path / "frame"
    out = subprocess.run([str(exe), str(base), os.path.join(ROOT, "models", "cornell_box.glb")], capture_output=True, text=True, check=True).stdout
    assert "160 x 90, sample 4, uploads 3" in out and "Image saved" in out
    png = np.asarray(Image.open(str(base) + ".png"))
    r = drt.Renderer(0)
    sc = drt.Scene(); sc.loadGLTFmodel(os.path.join(ROOT, "models", "cornell_box.glb"))
    b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8; b.buildIterative(sc)
    cam = drt.Camera((3.6, 1.25, 0.0)); cam.m_Forward_dir = np.float32([-1, 0, 0])
    r.ResizeBuffer(160, 90)
    for _ in range(3):
        r.Render(cam, sc)
    want8 = np.rint(np.clip(r.GetRenderTargetImage(), 0, 1) * np.float32(255)).astype(np.uint8)[::-1]
    assert np.array_equal(png, want8)


def test_isa_slot_table_behind_algorithmic_frac_is_reproducible(tmp_path):
    """profiles/r03_isa_slots.json -- the VALU issue slots of one call of each reference operation, which bench.py multiplies with
    the exact work counters to get roofline.algorithmic_frac -- is what tools/isa_by_phase.py --slots-json makes of
    tools/probes/isa_probes.hip with the Makefile's flags TODAY (no GPU needed: hipcc cross-compiles, llvm-objdump and
    llvm-symbolizer count).  A change of device_math.hpp that changes an operation's cost must come with a regenerated table."""
    import json, os, shutil, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc here")
    out = tmp_path / "slots.json"
    subprocess.run([sys.executable, os.path.join(root, "tools", "isa_by_phase.py"), "--slots-json", str(out)], check=True, capture_output=True, timeout=600)
    fresh = json.load(open(out))["per_call"]
    committed = json.load(open(os.path.join(root, "profiles", "r03_isa_slots.json")))["per_call"]
    assert fresh == committed
    assert 55 <= committed["tri_test"]["slots"] <= 70 and 20 <= committed["box_test"]["slots"] <= 30        # Moeller-Trumbore / slab test, unfused fp32
