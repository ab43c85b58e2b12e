"""Error behaviour of the C ABI (through the Python mirror): every misuse is a status code + message, never a crash or an exit
(the reference prints and calls exit(99), Editor/Common/CudaCommon.cu:4-13), and nothing falls back to a CPU path."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.scenes import scene_path

drt = pytest.importorskip("dustraytracer_amd")


def _code(fn, *args):
    rc = fn(*args)
    return rc, (drt._lib.drt_last_error() or b"").decode()


def test_scene_calls_reject_bad_arguments_without_a_gpu():
    lib = drt._lib
    sc = drt.Scene()
    assert _code(lib.drt_scene_load_gltf, None, b"x.glb")[0] == drt.ERR_INVALID
    assert _code(lib.drt_scene_load_gltf, sc._h, None)[0] == drt.ERR_INVALID
    rc, msg = _code(lib.drt_scene_load_gltf_ex, sc._h, b"x.glb", 0x80)
    assert rc == drt.ERR_INVALID and "flag" in msg
    assert _code(lib.drt_scene_set_geometry, sc._h, None, None, None, None, 3)[0] == drt.ERR_INVALID
    assert _code(lib.drt_scene_add_material, sc._h, None, -1)[0] == drt.ERR_INVALID
    tex = np.zeros((2, 2, 3), np.uint8)
    assert _code(lib.drt_scene_add_texture, sc._h, tex.ctypes.data, 2, 2, 7)[0] == drt.ERR_INVALID          # 7 channels
    assert _code(lib.drt_scene_add_texture, sc._h, tex.ctypes.data, 0, 2, 3)[0] == drt.ERR_INVALID
    info = drt._TexInfo()
    assert _code(lib.drt_scene_get_texture_info, sc._h, 0, C.byref(info))[0] == drt.ERR_INVALID            # no textures yet
    with pytest.raises(drt.DrtError) as e:                       # reference: m_BinCount = 8 (BVHBuilder.cuh:28); < 2 has no plane to try
        b = drt.BVHBuilder(); b.m_BinCount = 1; b.buildIterative(sc)
    assert e.value.code == drt.ERR_INVALID and "bin_count" in str(e.value)
    assert _code(lib.drt_scene_build_bvh, None, 8, 8)[0] == drt.ERR_INVALID
    with pytest.raises(drt.DrtError) as e:
        sc.loadGLTFmodel("/nonexistent/scene.glb")
    assert e.value.code == drt.ERR_IO
    sc.loadGLTFmodel(scene_path("room"))
    n = lib.drt_scene_triangle_count(sc._h)
    out = np.zeros(2, drt.TRIANGLE_DTYPE)
    assert _code(lib.drt_scene_get_triangles, sc._h, out.ctypes.data, -1)[0] == drt.ERR_INVALID
    assert _code(lib.drt_scene_get_triangles, sc._h, None, 5)[0] == drt.ERR_INVALID
    assert lib.drt_scene_get_triangles(sc._h, None, 0) == 0
    assert n > 2 and lib.drt_scene_get_triangles(sc._h, out.ctypes.data, 2) == 2      # a short destination is filled, not overrun
    sc.addTexture(np.zeros((4, 4, 3), np.uint8))
    assert lib.drt_scene_texture_count(sc._h) == 1
    assert lib.drt_scene_get_texture_info(sc._h, 0, C.byref(info)) == drt.OK
    tiny = np.zeros(4, np.uint8)
    rc, msg = _code(lib.drt_scene_get_texture_texels, sc._h, 0, tiny.ctypes.data, tiny.size)
    assert rc == drt.ERR_INVALID and "too small" in msg
    assert _code(lib.drt_scene_get_texture_texels, sc._h, 10 ** 6, tiny.ctypes.data, tiny.size)[0] == drt.ERR_INVALID
    # null handles never crash: counts of nothing are zero
    assert lib.drt_scene_triangle_count(None) == 0 and lib.drt_scene_bvh_depth(None) == 0
    assert lib.drt_renderer_width(None) == 0
    lib.drt_scene_destroy(None); lib.drt_renderer_destroy(None)
    assert lib.drt_shard_rows(1080, 0, 0, 8) == 0 and lib.drt_shard_rows(1080, 8, 8, 8) == 0
    assert _code(lib.drt_assemble_shards, None, None, 4, 4, 1, 1, 4, None)[0] == drt.ERR_INVALID
    assert _code(lib.drt_debug_kat, 9, 0, None, 0, None, 0, 0, None, 0, 0)[0] == drt.ERR_INVALID
    assert _code(lib.drt_debug_decode_image, b"GIF89a..", 8, C.byref(info), None, 0)[0] == drt.ERR_UNSUPPORTED


def test_indices_the_kernels_trust_are_checked_on_the_host():
    """drt_scene_set_geometry / _add_material accept any material id / texture index; the kernels index mats[] and texs[]
    unchecked, so pack() (run by drt_scene_validate and before every upload) refuses what would be an out-of-bounds GPU read."""
    def tri_scene(mat_ids, n_mats=1, tex=-1, n_tex=0):
        sc = drt.Scene()
        n = len(mat_ids)
        pos = np.tile(np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32), (n, 1)) + np.arange(n, dtype=np.float32)[:, None]
        nrm = np.tile(np.array([[0, 0, 1] * 3], np.float32), (n, 1))
        sc.setGeometry(pos, nrm, np.zeros((n, 6), np.float32), mat_ids)
        for _ in range(n_tex):
            sc.addTexture(np.zeros((2, 2, 3), np.uint8))
        for _ in range(n_mats):
            sc.addMaterial((0.5, 0.5, 0.5), tex)
        drt.BVHBuilder().buildIterative(sc)
        return sc
    tri_scene([0, 0, 0]).validate()
    tri_scene([0, 1], n_mats=2, tex=0, n_tex=1).validate()
    for bad in ([0, 1], [-1], [0, 7, 0]):
        with pytest.raises(drt.DrtError) as e:
            tri_scene(bad).validate()
        assert e.value.code == drt.ERR_INVALID and "material" in str(e.value)
    with pytest.raises(drt.DrtError) as e:
        tri_scene([0], tex=0).validate()                         # texture 0 of none
    assert e.value.code == drt.ERR_INVALID and "texture" in str(e.value)
    with pytest.raises(drt.DrtError) as e:
        tri_scene([0], tex=3, n_tex=2).validate()
    assert "texture" in str(e.value)
    sc = drt.Scene()
    with pytest.raises(ValueError):                              # the Python mirror checks the array lengths it passes on
        sc.setGeometry(np.zeros((2, 9), np.float32), np.zeros((2, 9), np.float32), np.zeros((2, 6), np.float32), [0, 0, 0])
    with pytest.raises(drt.DrtError) as e:
        drt.Scene().validate()                                   # no BVH
    assert "build_bvh" in str(e.value)


def test_without_a_gpu_the_renderer_refuses_to_exist():
    """No CPU fallback: on a machine without a usable HIP device creating a renderer is an error, not a slower path."""
    if drt.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(drt.DrtError) as e:
        drt.Renderer(0)
    assert e.value.code == drt.ERR_DEVICE
    assert drt._lib.drt_renderer_create(0) in (None, 0)
    sc = drt.Scene()
    sc.loadGLTFmodel(scene_path("room"))
    b = drt.BVHBuilder()
    b.m_BuildDevice = 0                                          # the GPU builder does not quietly run the host one either
    with pytest.raises(drt.DrtError) as e:
        b.buildIterative(sc)
    assert e.value.code == drt.ERR_DEVICE and len(sc.m_BVHNodes) == 0


@pytest.mark.gpu
def test_renderer_calls_reject_misuse():
    lib = drt._lib
    sc = drt.Scene()
    sc.loadGLTFmodel(scene_path("room"))
    cam = drt.Camera((0, 1.4, 2.0))
    r = drt.Renderer(0)
    r.ResizeBuffer(32, 32)
    with pytest.raises(drt.DrtError) as e:                       # no BVH yet
        r.Render(cam, sc)
    assert e.value.code == drt.ERR_INVALID and "build_bvh" in str(e.value)
    drt.BVHBuilder().buildIterative(sc)
    r2 = drt.Renderer(0)
    with pytest.raises(drt.DrtError) as e:                       # ResizeBuffer never called
        r2.Render(cam, sc)
    assert "ResizeBuffer" in str(e.value)
    with pytest.raises(drt.DrtError) as e:
        r2.ResizeBuffer(65536, 65536)                            # pixel index is 32-bit in the reference (RayGen.cuh:74)
    assert "too large" in str(e.value)
    with pytest.raises(drt.DrtError):
        r2.setShard(0, 0, 1)
    with pytest.raises(drt.DrtError):
        r2.setShard(8, 3, 2)                                     # rank >= world
    r2.ResizeBuffer(16, 16)
    small = np.zeros(10, np.float32)
    assert lib.drt_renderer_read_rgba32f(r2._h, small.ctypes.data, small.size) == drt.ERR_INVALID
    assert lib.drt_renderer_bind_buffers(r2._h, 1, None) == drt.ERR_INVALID          # both or neither
    assert lib.drt_renderer_set_frames_in_flight(r2._h, 0) == drt.ERR_INVALID
    assert lib.drt_renderer_render(None, None, None, None) == drt.ERR_INVALID
    # after all that the renderer still works
    r2.Render(cam, sc)
    assert r2.getSampleCount() == 2 and np.isfinite(r2.GetRenderTargetImage()).all()
    # a renderer on a device that does not exist
    with pytest.raises(drt.DrtError) as e:
        drt.Renderer(97)
    assert "out of range" in str(e.value)


def test_ring_protocol_of_the_pool_kernel_has_no_interleaving_that_puts_a_path_in_two_hands():
    """kernel_path_pool.hip's queues (multi-producer / multi-consumer rings in LDS): every interleaving of three waves on a
    ring of two and of four slots is explored.  Round 2's protocol (a consumer takes any non-empty slot) has a schedule in
    which one path id is taken twice; the lap-tagged entries shipped since round 3 have none (tools/sim_ring_protocol.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sim_ring_protocol", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "sim_ring_protocol.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    hazard, _ = m.explore("r02")
    assert hazard is not None and "already held" in hazard[-1][1]
    for kw in (dict(), dict(cap=4, n_ids=3), dict(cap=4, n_ids=4, waves=2, cycles=4)):
        clean, states = m.explore("lap", **kw)
        assert clean is None and states > 1000
