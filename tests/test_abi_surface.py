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
    assert drt._lib.drt_abi_version() == 1
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
