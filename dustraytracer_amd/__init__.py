"""dustraytracer_amd -- MI355X-native path-tracing core behind DustRayTracer's Renderer/Scene API.

Python host-side mirror of the reference's classes over the C ABI of include/drt.h
(dustraytracer_amd/libdrt_hip.so, hand-written HIP for gfx950).  Names follow the reference:

    Scene.loadGLTFmodel            Core/Scene/Scene.cuh:41-57
    BVHBuilder.buildIterative      Core/BVH/BVHBuilder.cuh:12-23
    Camera                         Core/Scene/Camera.cuh:14-48
    RendererSettings               Core/Scene/RendererSettings.h:4-35
    Renderer.ResizeBuffer/Render/resetAccumulationBuffer/getSampleCount   Core/Renderer.hpp:14-47

There is no CPU fallback: importing works anywhere the shared library loads, but creating a Renderer
without a GPU raises DrtError.  A missing shared library raises ImportError (build it with
`python -c "import __graft_entry__ as g; g.build()"` or `make -C dustraytracer_amd/csrc`).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DRT_LIB_OVERRIDE") or os.path.join(_HERE, "libdrt_hip.so")      # override: A/B of builds (tools/ab_libs.py)

if not os.path.exists(LIB_PATH):
    raise ImportError("dustraytracer_amd: %s is missing -- the HIP extension must be built "
                      "(make -C dustraytracer_amd/csrc); there is no fallback path" % LIB_PATH)


def _preload_hip_runtime():
    """One HIP runtime per process.  libdrt_hip.so needs libamdhip64.so.7; PyTorch-ROCm wheels bundle
    their own copy under the same soname.  Whichever copy is loaded first serves both, and torch
    finds no GPU when the system copy was loaded before its own.  So when a torch wheel is installed
    (and DRT_HIP_RUNTIME != "system") its copy is loaded first, without importing torch."""
    import importlib.util
    import sys
    if os.environ.get("DRT_HIP_RUNTIME", "torch") == "system" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


_preload_hip_runtime()
_lib = C.CDLL(LIB_PATH)


class DrtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("drt error %d: %s" % (code, message))
        self.code = code


OK, ERR_INVALID, ERR_IO, ERR_PARSE, ERR_UNSUPPORTED, ERR_DEVICE, ERR_BVH = 0, -1, -2, -3, -4, -5, -6


class RendererSettings(C.Structure):
    """Core/Scene/RendererSettings.h:4-35"""
    NORMALMODE, DEBUGMODE = 0, 1
    ALBEDO_DEBUG, NORMAL_DEBUG, BARYCENTRIC_DEBUG, UVS_DEBUG, MESHBVH_DEBUG, WORLDBVH_DEBUG = range(6)
    _fields_ = [("gamma_correction", C.c_int32), ("tone_mapping", C.c_int32), ("enableSunlight", C.c_int32),
                ("max_samples", C.c_int32), ("ray_bounce_limit", C.c_int32), ("RenderMode", C.c_int32),
                ("DebugMode", C.c_int32), ("sunlight_dir", C.c_float * 2), ("sunlight_color", C.c_float * 3),
                ("sunlight_intensity", C.c_float), ("sky_color", C.c_float * 3), ("sky_intensity", C.c_float)]

    def __init__(self, **kw):
        super().__init__()
        _lib.drt_default_settings(C.byref(self))
        for k, v in kw.items():
            _assign(self, k, v)


class MaterialModel(C.Structure):
    """include/drt.h drt_material_model: opt-in emissive term / metallic lobe (NOT reference behaviour; all zero = reference image)."""
    _fields_ = [("emissive", C.c_int32), ("specular", C.c_int32), ("emissive_scale", C.c_float), ("transmission", C.c_int32)]

    def __init__(self, emissive=0, specular=0, emissive_scale=1.0, transmission=0):
        super().__init__(int(emissive), int(specular), float(emissive_scale), int(transmission))


class _CameraPOD(C.Structure):
    _fields_ = [("exposure", C.c_float), ("vfov_rad", C.c_float), ("defocus_angle", C.c_float),
                ("focus_dist", C.c_float), ("position", C.c_float * 3), ("forward", C.c_float * 3)]


class Camera:
    """Core/Scene/Camera.cuh:14-48 (public fields + OnUpdate/Rotate/GetPosition)."""

    def __init__(self, pos=(0.0, 2.0, 5.0)):
        pod = _CameraPOD()
        _lib.drt_default_camera(C.byref(pod))
        self.exposure, self.vfov_rad = pod.exposure, pod.vfov_rad
        self.defocus_angle, self.focus_dist = pod.defocus_angle, pod.focus_dist
        self.m_movement_speed = 10.0
        self.m_Position = np.array(pos, np.float32)
        self.m_Forward_dir = np.array([0, 0, -1], np.float32)
        self.m_Up_dir = np.array([0, 1, 0], np.float32)
        self.m_Right_dir = np.cross(self.m_Forward_dir, self.m_Up_dir).astype(np.float32)

    def GetPosition(self):
        return self.m_Position.copy()

    def OnUpdate(self, velocity, delta):
        """Camera.cu:44-58: move along the camera basis (the C ABI's implementation: fp32, the reference's order)."""
        v = np.ascontiguousarray(velocity, np.float32)
        self.m_Position = np.ascontiguousarray(self.m_Position, np.float32)
        r, u, f = (np.ascontiguousarray(a, np.float32) for a in (self.m_Right_dir, self.m_Up_dir, self.m_Forward_dir))
        _lib.drt_camera_move(self.m_Position.ctypes.data, r.ctypes.data, u.ctypes.data, f.ctypes.data, v.ctypes.data,
                             C.c_float(self.m_movement_speed), C.c_float(delta))

    def Rotate(self, delta):
        """Camera.cu:61-80: delta = (sin_x, cos_x, sin_y, cos_y)."""
        d = np.ascontiguousarray(delta, np.float32)
        self.m_Forward_dir = np.ascontiguousarray(self.m_Forward_dir, np.float32)
        self.m_Right_dir = np.ascontiguousarray(self.m_Right_dir, np.float32)
        u = np.ascontiguousarray(self.m_Up_dir, np.float32)
        _lib.drt_camera_rotate(self.m_Forward_dir.ctypes.data, self.m_Right_dir.ctypes.data, u.ctypes.data, d.ctypes.data)

    def _pod(self):
        pod = _CameraPOD()
        pod.exposure, pod.vfov_rad = self.exposure, self.vfov_rad
        pod.defocus_angle, pod.focus_dist = self.defocus_angle, self.focus_dist
        for i in range(3):
            pod.position[i] = float(self.m_Position[i])
            pod.forward[i] = float(self.m_Forward_dir[i])
        return pod


TRIANGLE_DTYPE = np.dtype({"names": ["centroid", "vertex", "face_normal", "material"],
                           "formats": [("<f4", 3), (np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("uv", "<f4", 2)]), 3),
                                       ("<f4", 3), "<i4"],
                           "offsets": [0, 16, 112, 124], "itemsize": 128})
NODE_DTYPE = np.dtype({"names": ["is_leaf", "bmin", "bmax", "child1", "child2", "prim_count", "prim_start"],
                       "formats": ["u1", ("<f4", 3), ("<f4", 3), "<i4", "<i4", "<i4", "<i4"],
                       "offsets": [0, 4, 16, 28, 32, 36, 40], "itemsize": 44})
MATERIAL_DTYPE = np.dtype({"names": ["albedo", "emissive", "albedo_tex", "roughness", "transmission", "refractive_index", "metallic"],
                           "formats": [("<f4", 3), ("<f4", 3), "<i4", "<f4", "u1", "<f4", "u1"],
                           "offsets": [0, 12, 24, 28, 32, 36, 40], "itemsize": 44})
MESH_DTYPE = np.dtype([("primitives_offset", "<i4"), ("tris_count", "<i4")])


class _TexInfo(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("components", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "rays", "node_visits", "inner_visits", "tri_tests",
                                           "hits_textured", "hits_flat", "shadow_rays", "inner_visits_shadow",
                                           "tri_tests_shadow")] + [("phase_execs", C.c_uint64 * 4), ("phase_lanes", C.c_uint64 * 4), ("phase_ticks", C.c_uint64 * 4), ("wave_ticks", C.c_uint64), ("sampler_tries", C.c_uint64)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, t in self._fields_ if t is C.c_uint64 and n != "wave_ticks"}

    def phase_stats(self):
        """wave_queue kernel: (executions, mean active lanes) of the T, N, S, R phases."""
        return {k: (int(self.phase_execs[i]), self.phase_lanes[i] / max(int(self.phase_execs[i]), 1))
                for i, k in enumerate("TNSR")}

    def algorithmic_bytes(self):
        """40 B/sample + 56 B/interior visit + 36 B/triangle test + 60|32 B/shaded hit (+ shadow terms)."""
        return (40 * self.samples + 56 * self.inner_visits + 36 * self.tri_tests + 60 * self.hits_textured
                + 32 * self.hits_flat + 56 * self.inner_visits_shadow + 36 * self.tri_tests_shadow)


def _sig(name, restype, *argtypes):
    if name.startswith("drt_debug_") and not hasattr(_lib, name):
        return None                   # older A/B builds (DRT_LIB_OVERRIDE) may lack a debug entry point
    fn = getattr(_lib, name)          # AttributeError here = the library does not export what drt.h declares
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


_P = C.c_void_p
_sig("drt_abi_version", C.c_int)
_sig("drt_last_error", C.c_char_p)
_sig("drt_device_count", C.c_int)
_sig("drt_default_settings", None, C.POINTER(RendererSettings))
_sig("drt_default_camera", None, C.POINTER(_CameraPOD))
_sig("drt_camera_rotate", None, _P, _P, _P, _P)
_sig("drt_camera_move", None, _P, _P, _P, _P, _P, C.c_float, C.c_float)
_sig("drt_scene_create", _P)
_sig("drt_scene_destroy", None, _P)
_sig("drt_scene_load_gltf", C.c_int, _P, C.c_char_p)
_sig("drt_scene_load_gltf_ex", C.c_int, _P, C.c_char_p, C.c_uint32)
_sig("drt_scene_set_geometry", C.c_int, _P, _P, _P, _P, _P, C.c_int32)
_sig("drt_scene_add_material", C.c_int, _P, C.POINTER(C.c_float), C.c_int32)
_sig("drt_scene_add_texture", C.c_int, _P, _P, C.c_int32, C.c_int32, C.c_int32)
_sig("drt_scene_build_bvh", C.c_int, _P, C.c_int32, C.c_int32)
_sig("drt_scene_validate", C.c_int, _P)
_sig("drt_scene_add_material_ex", C.c_int, _P, _P)
_sig("drt_pcg_hash", C.c_uint32, C.c_uint32)
_sig("drt_random_float", C.c_float, C.POINTER(C.c_uint32))
_sig("drt_scene_build_bvh_recursive", C.c_int, _P, C.c_int32, C.c_int32)
_sig("drt_scene_build_bvh_device", C.c_int, _P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float))
for _n in ("triangle", "node", "material", "texture", "mesh"):
    _sig("drt_scene_%s_count" % _n, C.c_int32, _P)
_sig("drt_scene_bvh_depth", C.c_int32, _P)
for _n in ("triangles", "nodes", "materials", "meshes"):
    _sig("drt_scene_get_%s" % _n, C.c_int, _P, _P, C.c_int32)
_sig("drt_scene_get_texture_info", C.c_int, _P, C.c_int32, C.POINTER(_TexInfo))
_sig("drt_scene_get_texture_texels", C.c_int, _P, C.c_int32, _P, C.c_size_t)
_sig("drt_renderer_create", _P, C.c_int32)
_sig("drt_renderer_destroy", None, _P)
_sig("drt_renderer_resize", C.c_int, _P, C.c_uint32, C.c_uint32)
_sig("drt_renderer_set_settings", C.c_int, _P, C.POINTER(RendererSettings))
_sig("drt_renderer_get_settings", C.c_int, _P, C.POINTER(RendererSettings))
_sig("drt_renderer_render", C.c_int, _P, C.POINTER(_CameraPOD), _P, C.POINTER(C.c_float))
_sig("drt_renderer_render_batch", C.c_int, _P, C.POINTER(_CameraPOD), _P, C.c_uint32, C.POINTER(C.c_float))
_sig("drt_renderer_render_batch_async", C.c_int, _P, C.POINTER(_CameraPOD), _P, C.c_uint32)
_sig("drt_renderer_wait", C.c_int, _P, C.POINTER(C.c_float))
_sig("drt_renderer_reset", C.c_int, _P)
for _n in ("width", "height", "sample_count", "local_rows"):
    _sig("drt_renderer_%s" % _n, C.c_uint32, _P)
_sig("drt_renderer_read_rgba32f", C.c_int, _P, _P, C.c_size_t)
_sig("drt_renderer_read_accum", C.c_int, _P, _P, C.c_size_t)
_sig("drt_renderer_device_rgba", _P, _P)
_sig("drt_renderer_device_accum", _P, _P)
_sig("drt_renderer_set_shard", C.c_int, _P, C.c_uint32, C.c_uint32, C.c_uint32)
_sig("drt_renderer_bind_buffers", C.c_int, _P, _P, _P)
_sig("drt_renderer_set_stream", C.c_int, _P, _P)
_sig("drt_renderer_set_counting", C.c_int, _P, C.c_int32)
_sig("drt_renderer_get_counters", C.c_int, _P, C.POINTER(Counters))
_sig("drt_renderer_set_material_model", C.c_int, _P, _P)
_sig("drt_renderer_get_material_model", C.c_int, _P, _P)
_sig("drt_renderer_kernel_info", C.c_int, _P, C.c_char_p, C.c_size_t)
_sig("drt_renderer_kernel_span", C.c_int, _P, C.POINTER(C.c_float))
_sig("drt_renderer_launch_count", C.c_int32, _P)
_sig("drt_renderer_set_frames_in_flight", C.c_int, _P, C.c_int32)
_sig("drt_assemble_shards", C.c_int, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P)
_sig("drt_shard_rows", C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32)
_sig("drt_debug_decode_image", C.c_int, _P, C.c_size_t, _P, _P, C.c_size_t)
_sig("drt_debug_kat", C.c_int, C.c_int32, C.c_int32, _P, C.c_size_t, _P, C.c_size_t, C.c_uint32, C.POINTER(_CameraPOD), C.c_uint32, C.c_uint32)
_sig("drt_debug_hash_cycles", C.c_int, C.c_int32, C.c_uint32, _P, C.c_uint32, C.POINTER(C.c_uint32))
_sig("drt_group_create", _P, C.POINTER(C.c_int32), C.c_int32)
_sig("drt_group_destroy", None, _P)
_sig("drt_group_size", C.c_int32, _P)
_sig("drt_group_renderer", _P, _P, C.c_int32)
_sig("drt_group_resize", C.c_int, _P, C.c_uint32, C.c_uint32)
_sig("drt_group_set_settings", C.c_int, _P, _P)
_sig("drt_group_reset", C.c_int, _P)
_sig("drt_group_render_batch", C.c_int, _P, _P, _P, C.c_uint32, C.POINTER(C.c_float))
_sig("drt_group_render_batch_async", C.c_int, _P, _P, _P, C.c_uint32)
_sig("drt_group_wait", C.c_int, _P, C.POINTER(C.c_float))
_sig("drt_group_sample_count", C.c_uint32, _P)
_sig("drt_group_device_rgba", _P, _P)
_sig("drt_group_read_rgba32f", C.c_int, _P, _P, C.c_size_t)
_sig("drt_shard_stripe", C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
_sig("drt_debug_wave_queue_plans", C.c_int, _P, C.c_char_p, C.c_size_t)
_sig("drt_debug_pool_stats", C.c_int, _P, _P, C.c_int32)
_sig("drt_debug_check_rcp", C.c_int, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
_sig("drt_debug_check_sqrt", C.c_int, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))

EXPORTED_SYMBOLS = [n for n in dir(_lib) if n.startswith("drt_")]


def _check(rc):
    if rc < 0:
        raise DrtError(rc, (_lib.drt_last_error() or b"").decode("utf-8", "replace"))
    return rc


def _assign(struct, key, value):
    cur = getattr(struct, key)
    if hasattr(cur, "__len__"):
        for i, v in enumerate(value):
            cur[i] = v
    else:
        setattr(struct, key, value)


def device_count():
    return _lib.drt_device_count()


def shard_rows(height, stripe_rows, rank, world):
    return int(_lib.drt_shard_rows(height, stripe_rows, rank, world))


class Scene:
    """Core/Scene/Scene.cuh:41-57: loadGLTFmodel + the public buffers (as numpy copies)."""

    def __init__(self):
        self._h = _lib.drt_scene_create()
        if not self._h:
            raise DrtError(ERR_INVALID, "cannot create scene")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:              # (at interpreter shutdown the module's globals may be gone already)
            _lib.drt_scene_destroy(h)

    def loadGLTFmodel(self, filepath, strict=False):
        """strict=False: the reference's reading of the file (Scene.cu, quirks included); True: the glTF 2.0 specification's
        (node transforms and hierarchy, accessor offsets / strides / component types, u8/u16/u32 or no indices, ...)."""
        _check(_lib.drt_scene_load_gltf_ex(self._h, os.fsencode(filepath), 1 if strict else 0))
        return True

    def setGeometry(self, positions, normals, uvs, material_ids):
        pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 9)
        nrm = np.ascontiguousarray(normals, np.float32).reshape(-1, 9)
        uv = np.ascontiguousarray(uvs, np.float32).reshape(-1, 6)
        mat = np.ascontiguousarray(material_ids, np.int32).reshape(-1)
        if not (len(pos) == len(nrm) == len(uv) == len(mat)):
            raise ValueError("setGeometry: %d position, %d normal, %d uv triangles for %d material ids" % (len(pos), len(nrm), len(uv), len(mat)))
        _check(_lib.drt_scene_set_geometry(self._h, pos.ctypes.data, nrm.ctypes.data, uv.ctypes.data, mat.ctypes.data, len(mat)))

    def validate(self):
        """Raises DrtError(ERR_INVALID) if a triangle names a material, or a material a texture, that does not exist."""
        _check(_lib.drt_scene_validate(self._h))

    def addMaterial(self, albedo, albedo_tex=-1):
        a = (C.c_float * 3)(*albedo)
        return _check(_lib.drt_scene_add_material(self._h, a, albedo_tex))

    def addMaterialEx(self, albedo, albedo_tex=-1, emissive=(0, 0, 0), roughness=0.0, metallic=False, transmission=False, refractive_index=1.45):
        """A material with the fields only the opt-in material model reads (Renderer.setMaterialModel)."""
        m = np.zeros(1, MATERIAL_DTYPE)
        m["albedo"], m["emissive"], m["albedo_tex"], m["roughness"], m["metallic"] = albedo, emissive, albedo_tex, roughness, int(bool(metallic))
        m["transmission"], m["refractive_index"] = int(bool(transmission)), refractive_index
        return _check(_lib.drt_scene_add_material_ex(self._h, m.ctypes.data))

    def addTexture(self, texels):
        t = np.ascontiguousarray(texels, np.uint8)
        h, w, c = t.shape
        return _check(_lib.drt_scene_add_texture(self._h, t.ctypes.data, w, h, c))

    def _copy(self, getter, count, dtype):
        out = np.zeros(count, dtype)
        if count:
            _check(getter(self._h, out.ctypes.data, count))
        return out

    @property
    def m_PrimitivesBuffer(self):
        return self._copy(_lib.drt_scene_get_triangles, _lib.drt_scene_triangle_count(self._h), TRIANGLE_DTYPE)

    @property
    def m_BVHNodes(self):
        return self._copy(_lib.drt_scene_get_nodes, _lib.drt_scene_node_count(self._h), NODE_DTYPE)

    @property
    def m_Material(self):
        return self._copy(_lib.drt_scene_get_materials, _lib.drt_scene_material_count(self._h), MATERIAL_DTYPE)

    @property
    def m_Meshes(self):
        return self._copy(_lib.drt_scene_get_meshes, _lib.drt_scene_mesh_count(self._h), MESH_DTYPE)

    @property
    def m_Textures(self):
        out = []
        for i in range(_lib.drt_scene_texture_count(self._h)):
            info = _TexInfo()
            _check(_lib.drt_scene_get_texture_info(self._h, i, C.byref(info)))
            tex = np.zeros((info.height, info.width, info.components), np.uint8)
            _check(_lib.drt_scene_get_texture_texels(self._h, i, tex.ctypes.data, tex.size))
            out.append(tex)
        return out

    @property
    def bvh_depth(self):
        return _lib.drt_scene_bvh_depth(self._h)


class BVHBuilder:
    """Core/BVH/BVHBuilder.cuh:12-23 (defaults of the class; the editor sets 20 / 8, EditorLayer.cpp:52-55)."""

    def __init__(self):
        self.m_BinCount = 8
        self.m_TargetLeafPrimitivesCount = 6
        self.m_BuildDevice = -1          # new: >= 0 builds the same tree on that GPU (drt_scene_build_bvh_device)
        self.m_LastBuildDeviceMs = 0.0

    def buildIterative(self, scene):
        if self.m_BuildDevice >= 0:
            ms = C.c_float(0)
            _check(_lib.drt_scene_build_bvh_device(scene._h, self.m_TargetLeafPrimitivesCount, self.m_BinCount, self.m_BuildDevice, C.byref(ms)))
            self.m_LastBuildDeviceMs = ms.value
        else:
            _check(_lib.drt_scene_build_bvh(scene._h, self.m_TargetLeafPrimitivesCount, self.m_BinCount))
        return scene

    def build(self, scene):
        """BVHBuilder::build (BVHBuilder.cu:100-173): the same tree and triangle order through recursion; the node array comes in
        the recursion's order (children after both of their subtrees, root last), as drt_scene_get_nodes then returns it."""
        _check(_lib.drt_scene_build_bvh_recursive(scene._h, self.m_TargetLeafPrimitivesCount, self.m_BinCount))
        return scene


def shard_stripe(width, height, stripe_rows, rank, world, k):
    """(offset in the rank's compact shard, offset in the full image, length) of the rank's k-th stripe, in floats of an
    RGBA32F frame; None when the rank has no k-th stripe."""
    so, do, cnt = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    if not _lib.drt_shard_stripe(width, height, stripe_rows, rank, world, k, C.byref(so), C.byref(do), C.byref(cnt)):
        return None
    return int(so.value), int(do.value), int(cnt.value)


class RendererGroup:
    """Several GPUs of one node behind the Renderer interface (drt_group_*): one process, a renderer per device, stripes
    gathered into the first device's image over RCCL.  Same calls as Renderer."""

    def __init__(self, devices=(0,)):
        arr = (C.c_int32 * len(devices))(*devices)
        self._h = _lib.drt_group_create(arr, len(devices))
        if not self._h:
            raise DrtError(ERR_DEVICE, (_lib.drt_last_error() or b"").decode())
        self.m_RendererSettings = RendererSettings()
        self._w = self._h_px = 0

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:              # (at interpreter shutdown the module's globals may be gone already)
            _lib.drt_group_destroy(h)

    def size(self):
        return int(_lib.drt_group_size(self._h))

    def ResizeBuffer(self, width, height):
        _check(_lib.drt_group_resize(self._h, width, height))
        self._w, self._h_px = width, height

    def resetAccumulationBuffer(self):
        _check(_lib.drt_group_reset(self._h))

    def getSampleCount(self):
        return int(_lib.drt_group_sample_count(self._h))

    def RenderBatch(self, cam, scene, n_frames):
        _check(_lib.drt_group_set_settings(self._h, C.byref(self.m_RendererSettings)))
        ms = C.c_float(0)
        pod = cam._pod()
        _check(_lib.drt_group_render_batch(self._h, C.byref(pod), scene._h, int(n_frames), C.byref(ms)))
        return ms.value

    def Render(self, cam, scene):
        return self.RenderBatch(cam, scene, 1)

    def RenderBatchAsync(self, cam, scene, n_frames):
        _check(_lib.drt_group_set_settings(self._h, C.byref(self.m_RendererSettings)))
        pod = cam._pod()
        _check(_lib.drt_group_render_batch_async(self._h, C.byref(pod), scene._h, int(n_frames)))

    def Wait(self):
        ms = C.c_float(0)
        _check(_lib.drt_group_wait(self._h, C.byref(ms)))
        return ms.value

    def GetRenderTargetImage(self):
        out = np.zeros((self._h_px, self._w, 4), np.float32)
        _check(_lib.drt_group_read_rgba32f(self._h, out.ctypes.data, out.size))
        return out

    def kernelInfo(self, index=0):
        buf = C.create_string_buffer(128)
        _check(_lib.drt_renderer_kernel_info(_lib.drt_group_renderer(self._h, index), buf, 128))
        return buf.value.decode()


class Renderer:
    """Core/Renderer.hpp:14-47."""

    def __init__(self, device=0):
        self._h = _lib.drt_renderer_create(device)
        if not self._h:
            raise DrtError(ERR_DEVICE, (_lib.drt_last_error() or b"").decode())
        self.m_RendererSettings = RendererSettings()

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:              # (at interpreter shutdown the module's globals may be gone already)
            _lib.drt_renderer_destroy(h)

    def ResizeBuffer(self, width, height):
        _check(_lib.drt_renderer_resize(self._h, width, height))

    def _push_settings(self):
        _check(_lib.drt_renderer_set_settings(self._h, C.byref(self.m_RendererSettings)))

    def setMaterialModel(self, emissive=0, specular=0, emissive_scale=1.0, transmission=0):
        """Opt-in extension (drt_material_model): emissive term, metallic lobe, dielectric lobe.  Off (the default) = the reference's image."""
        m = MaterialModel(emissive, specular, emissive_scale, transmission)
        _check(_lib.drt_renderer_set_material_model(self._h, C.byref(m)))

    def Render(self, cam, scene):
        """One frame index; returns the kernel time in ms (the reference's `float* delta`)."""
        self._push_settings()
        ms = C.c_float(0)
        pod = cam._pod()
        _check(_lib.drt_renderer_render(self._h, C.byref(pod), scene._h, C.byref(ms)))
        return ms.value

    def RenderBatch(self, cam, scene, n_frames):
        self._push_settings()
        ms = C.c_float(0)
        pod = cam._pod()
        _check(_lib.drt_renderer_render_batch(self._h, C.byref(pod), scene._h, n_frames, C.byref(ms)))
        return ms.value

    def RenderBatchAsync(self, cam, scene, n_frames):
        """Enqueue only; Wait() blocks and returns the device time in ms."""
        self._push_settings()
        pod = cam._pod()
        _check(_lib.drt_renderer_render_batch_async(self._h, C.byref(pod), scene._h, n_frames))

    def Wait(self):
        ms = C.c_float(0)
        _check(_lib.drt_renderer_wait(self._h, C.byref(ms)))
        return ms.value

    def resetAccumulationBuffer(self):
        _check(_lib.drt_renderer_reset(self._h))

    def getBufferWidth(self):
        return _lib.drt_renderer_width(self._h)

    def getBufferHeight(self):
        return _lib.drt_renderer_height(self._h)

    def getSampleCount(self):
        return _lib.drt_renderer_sample_count(self._h)

    def getLocalRows(self):
        return _lib.drt_renderer_local_rows(self._h)

    def GetRenderTargetImage(self):
        """RGBA32F framebuffer as numpy [local_rows, width, 4]; row 0 = bottom (replaces the GL texture name)."""
        out = np.zeros((self.getLocalRows(), self.getBufferWidth(), 4), np.float32)
        _check(_lib.drt_renderer_read_rgba32f(self._h, out.ctypes.data, out.size))
        return out

    def GetAccumulationBuffer(self):
        out = np.zeros((self.getLocalRows(), self.getBufferWidth(), 3), np.float32)
        _check(_lib.drt_renderer_read_accum(self._h, out.ctypes.data, out.size))
        return out

    def setShard(self, stripe_rows, rank, world):
        _check(_lib.drt_renderer_set_shard(self._h, stripe_rows, rank, world))

    def bindBuffers(self, accum_ptr, rgba_ptr):
        _check(_lib.drt_renderer_bind_buffers(self._h, accum_ptr, rgba_ptr))

    def setStream(self, stream_ptr):
        _check(_lib.drt_renderer_set_stream(self._h, stream_ptr))

    def setCounting(self, enable):
        _check(_lib.drt_renderer_set_counting(self._h, 1 if enable else 0))

    def getCounters(self):
        c = Counters()
        _check(_lib.drt_renderer_get_counters(self._h, C.byref(c)))
        return c

    def setFramesInFlight(self, n):
        """Hint that n launches are kept in flight on this device (other renderers on other streams): small launches get
        smaller grids so that they overlap.  Does not change results."""
        _check(_lib.drt_renderer_set_frames_in_flight(self._h, int(n)))

    def waveQueuePlans(self):
        """The launch packagings of wave_queue timed so far (list of plans; see drt_debug_wave_queue_plans)."""
        import json
        buf = C.create_string_buffer(1 << 16)
        _check(_lib.drt_debug_wave_queue_plans(self._h, buf, len(buf)))
        return json.loads(buf.value.decode())

    def poolStats(self, reset=True):
        """path_pool statistics (renderer created with DRT_POOL_STATS=1): dict queue -> (batches, mean paths per batch, ticks)."""
        a = np.zeros(40, np.uint64)
        _check(_lib.drt_debug_pool_stats(self._h, a.ctypes.data, 1 if reset else 0))
        out = {}
        for k, name in enumerate(("N", "T0", "T1", "T2", "T3", "B", "E", "R", "S")):
            b, l, t = int(a[3 * k]), int(a[3 * k + 1]), int(a[3 * k + 2])
            out[name] = (b, l / max(b, 1), t)
        out["claim_ticks"], out["idle_polls"], out["lost_claims"], out["wave_ticks"] = int(a[27]), int(a[28]), int(a[29]), int(a[30])
        out["failed_claims"], out["failed_claim_ticks"], out["idle_ticks"] = int(a[31]), int(a[32]), int(a[33])
        out["work"] = dict(n_iterations=int(a[34]), n_lane_pops=int(a[35]), t_steps=int(a[36]), t_lane_tests=int(a[37]), dir_iterations=int(a[38]), dir_lane_tries=int(a[39]))
        return out

    def launchesOfLastBatch(self):
        """Tracing-kernel launches the last RenderBatch was split into (per-launch sample buffer budget)."""
        return int(_lib.drt_renderer_launch_count(self._h))

    def kernelSpanMs(self):
        """Device-measured execution time of the tracing kernel(s) of the last completed batch (no queueing time)."""
        ms = C.c_float(0)
        _check(_lib.drt_renderer_kernel_span(self._h, C.byref(ms)))
        return float(ms.value)

    def kernelInfo(self):
        buf = C.create_string_buffer(128)
        _check(_lib.drt_renderer_kernel_info(self._h, buf, 128))
        return buf.value.decode()


_KAT_WORDS = {0: (1, 5), 1: (1, 5), 2: (12, 1), 3: (15, 5), 4: (3, 7), 5: (1, 3), 6: (10, 7)}


def debug_kat(which, inputs, cam=None, width=0, height=0, device=0):
    """Runs device leaf function `which` (see drt.h drt_debug_kat) on uint32-viewed inputs [n, words]; returns uint32 [n, words]."""
    win, wout = _KAT_WORDS[which]
    a = np.ascontiguousarray(inputs).view(np.uint32).reshape(-1, win)
    out = np.zeros((len(a), wout), np.uint32)
    pod = cam._pod() if cam is not None else None
    _check(_lib.drt_debug_kat(device, which, a.ctypes.data, a.nbytes, out.ctypes.data, out.nbytes, len(a),
                              C.byref(pod) if pod is not None else None, width, height))
    return out


def debug_decode_image(file_bytes):
    """The loader's image decoder (PNG / baseline JPEG) on a file held in memory -> uint8 [H, W, C]."""
    buf = (C.c_uint8 * len(file_bytes)).from_buffer_copy(bytes(file_bytes))
    info = _TexInfo()
    _check(_lib.drt_debug_decode_image(buf, len(file_bytes), C.byref(info), None, 0))
    out = np.zeros((info.height, info.width, info.components), np.uint8)
    _check(_lib.drt_debug_decode_image(buf, len(file_bytes), C.byref(info), out.ctypes.data, out.size))
    return out


def debug_hash_cycles(max_len=64, cap=4096, device=0):
    """[(value, cycle length)] for every 32-bit value on a pcg_hash cycle of length <= max_len."""
    pairs = np.zeros((cap, 2), np.uint32)
    found = C.c_uint32(0)
    _check(_lib.drt_debug_hash_cycles(device, max_len, pairs.ctypes.data, cap, C.byref(found)))
    return [(int(v), int(n)) for v, n in pairs[: min(found.value, cap)]]


def debug_check_rcp(device=0):
    """(mismatches, fast-path count) of the kernels' exact_rcp vs IEEE 1.0f/x over all 2^32 floats."""
    bad, fast = C.c_uint64(0), C.c_uint64(0)
    _check(_lib.drt_debug_check_rcp(device, C.byref(bad), C.byref(fast)))
    return int(bad.value), int(fast.value)


def debug_check_sqrt(device=0):
    """(mismatches, fast-path count) of the kernels' exact_sqrt vs sqrtf over all 2^32 floats."""
    bad, fast = C.c_uint64(0), C.c_uint64(0)
    _check(_lib.drt_debug_check_sqrt(device, C.byref(bad), C.byref(fast)))
    return int(bad.value), int(fast.value)


def assemble_shards(gathered_ptr, image_ptr, width, height, stripe_rows, world, padded_rows, stream_ptr=None):
    _check(_lib.drt_assemble_shards(gathered_ptr, image_ptr, width, height, stripe_rows, world, padded_rows, stream_ptr))
