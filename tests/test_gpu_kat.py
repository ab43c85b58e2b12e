"""-m gpu: the DEVICE leaf functions (RNG, slab, triangle test, camera ray) against tests/golden/kat_ref.npz,
i.e. against outputs of the reference's own compiled sources (generator: tests/golden/make_kat_golden.py).
Bit-exact.  Also the RNG cycle guard, which only exists because pcg_hash has short cycles."""
import numpy as np
import pytest

import oracle
from tests.scenes import bits

pytestmark = pytest.mark.gpu
drt = pytest.importorskip("dustraytracer_amd")


def test_device_unit_vectors_match_reference(kat_golden):
    g = kat_golden
    out = drt.debug_kat(0, g["seeds"])
    assert np.array_equal(out[:, :3], bits(g["unitvec"]))
    assert np.array_equal(out[:, 3], g["unitvec_seed"])
    out = drt.debug_kat(1, g["seeds"])
    assert np.array_equal(out[:, :3], bits(g["unitsphere"]))          # incl. the sqrt-free acceptance test
    assert np.array_equal(out[:, 3], g["unitsphere_seed"])
    out = drt.debug_kat(5, g["seeds"])
    assert np.array_equal(out[:, :2], bits(g["unitdisk"]))
    assert np.array_equal(out[:, 2], g["unitdisk_seed"])


def test_device_sphere_sampler_wide(kat_golden):
    """200k more seeds, against the oracle (itself pinned to the reference on the golden set)."""
    rng = np.random.default_rng(11)
    seeds = rng.integers(0, 2 ** 32, 200_000, dtype=np.uint64).astype(np.uint32)
    v, s, it = oracle.kat_unitsphere(seeds)
    out = drt.debug_kat(1, seeds)
    assert np.array_equal(out[:, :3], bits(v)) and np.array_equal(out[:, 3], s)
    assert np.array_equal(out[:, 4].astype(np.int64), it.astype(np.int64))


def test_device_slab_matches_reference(kat_golden):
    g = kat_golden
    out = drt.debug_kat(2, np.concatenate([g["slab_rays"], g["slab_boxes"]], axis=1).astype(np.float32))
    assert np.array_equal(out[:, 0], bits(g["slab"]))


def test_device_triangle_test_matches_reference(kat_golden):
    g = kat_golden
    out = drt.debug_kat(3, np.concatenate([g["isect_rays"], g["isect_tris"]], axis=1).astype(np.float32))
    hit = out[:, 4].astype(bool)
    assert np.array_equal(hit, g["isect_hit"].astype(bool))
    assert np.array_equal(out[hit, :4], bits(g["isect_tuvw"])[hit])      # t, U, V, W of the straight-line test + exact_rcp


def test_device_closest_hit_frame_matches_reference(kat_golden):
    g = kat_golden
    inp = np.concatenate([g["ch_rays"], g["ch_t"][:, None], g["ch_fn"]], axis=1).astype(np.float32)
    out = drt.debug_kat(6, inp)
    assert np.array_equal(out[:, :3], bits(g["ch_pos"]))
    assert np.array_equal(out[:, 3:6], bits(g["ch_normal"]))
    assert np.array_equal(out[:, 6].astype(np.int32), g["ch_front"])


def test_device_camera_ray_matches_reference(kat_golden):
    g = kat_golden
    for k, c in enumerate(g["cams"]):
        cam = drt.Camera([float(v) for v in c[4:7]])
        cam.exposure, cam.vfov_rad, cam.defocus_angle, cam.focus_dist = (float(c[0]), float(c[1]), float(c[2]), float(c[3]))
        cam.m_Forward_dir = np.array(c[7:10], np.float32)
        inp = np.zeros((len(g["seeds"]), 3), np.uint32)
        inp[:, :2] = bits(g["uv"])
        inp[:, 2] = g["seeds"]
        out = drt.debug_kat(4, inp, cam=cam, width=int(c[10]), height=int(c[11]))
        assert np.array_equal(out[:, 6], g["getray_seed"][k]), "camera %d seed stream" % k
        assert np.array_equal(out[:, :6], bits(g["getray"][k])), "camera %d" % k


def test_rng_hash_has_short_cycles_and_the_guard_terminates():
    """pcg_hash (Random.cu:6-11) is a permutation with short cycles; on some, every candidate of the rejection
    sampler is rejected and the reference's loop (Random.cu:50-58) never ends.  Kernel and oracle stop after 1024
    candidates and agree on the result."""
    cycles = drt.debug_hash_cycles(max_len=64)
    lengths = sorted({n for _, n in cycles})
    assert lengths[:4] == [4, 8, 10, 13]
    by_len = {}
    for v, n in cycles:
        by_len.setdefault(n, []).append(v)
    assert all(len(v) == n for n, v in by_len.items())                # a cycle of length n has n members
    seeds = np.array([v for v, _ in cycles], np.uint32)
    # check the cycle property with the oracle's hash
    for v, n in cycles[:8]:
        x = v
        for _ in range(n):
            x = int(oracle.kat_pcg([x])[0])
        assert x == v
    vec, s_out, tries = oracle.kat_unitsphere(seeds)
    out = drt.debug_kat(1, seeds)
    assert np.array_equal(out[:, :3], bits(vec)) and np.array_equal(out[:, 3], s_out)
    assert np.array_equal(out[:, 4].astype(np.int64), tries.astype(np.int64))
    assert tries.max() == 1024, "expected at least one start on a short cycle that never accepts (the reference would hang)"
    stuck = tries == 1024
    assert np.isfinite(vec[stuck]).all() and np.allclose(np.linalg.norm(vec[stuck].astype(np.float64), axis=1), 1.0, atol=1e-6)


def test_pixel_walk_cross_check_library():
    """Round 1's first kernel (one lane per pixel, nested loops) is no longer part of libdrt_hip.so; `make pixel-walk` builds
    dustraytracer_amd/libdrt_hip_pixel_walk.so with it, and a child process loads THAT library and asks it the same two
    questions as the production kernels: the hand-derived tie scene below, and a whole image against the oracle."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "dustraytracer_amd", "libdrt_hip_pixel_walk.so")
    assert os.path.exists(lib), "build it: make -C dustraytracer_amd/csrc pixel-walk (__graft_entry__.build() does)"
    with pytest.raises(drt.DrtError):                       # the shipped library refuses the name
        from tests.test_gpu_parity import _renderer_with_env
        _renderer_with_env({"DRT_KERNEL": "pixel_walk"})
    env = dict(os.environ, DRT_LIB_OVERRIDE=lib, DRT_PIXEL_WALK_CHILD="1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.abspath(__file__), "-k", "pixel_walk_child"], env=env, cwd=root,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.skipif(__import__("os").environ.get("DRT_PIXEL_WALK_CHILD") != "1", reason="runs in the child process of test_pixel_walk_cross_check_library")
def test_pixel_walk_child():
    from tests.test_gpu_parity import _render_with_env, compare
    test_equal_distances_and_equal_hits_resolve_as_in_the_reference_on_the_device("pixel_walk")
    r, ref, _ = _render_with_env({"DRT_KERNEL": "pixel_walk"}, "suzanne_plane", 128, 72, 2, 2)
    assert r.kernelInfo().startswith("pixel_walk")
    compare(r.GetRenderTargetImage(), ref, "pixel_walk")


@pytest.mark.parametrize("kernel", ["path_pool", "wave_queue"])
def test_equal_distances_and_equal_hits_resolve_as_in_the_reference_on_the_device(kernel):
    """Device side of the hand-derived traversal KAT (tests/test_oracle_kat.py tie_scene): child boxes entered at the same
    distance, triangles hit at the same t; BVHTraversal.cuh:51,63-70 make the triangle of child 1 win.  A camera with a zero
    field of view sends the same exact ray through every pixel, whatever the jitter.  Debug view (general wave_queue kernel /
    pixel_walk) shows the winner's albedo; the lean kernels (path_pool, wave_queue lean) show it through one sky bounce."""
    import os
    import oracle
    from tests.test_oracle_kat import tie_scene, tie_winner_albedo
    from tests.test_gpu_parity import _programmatic_pair, compare
    pos, nrm, uv, mat, materials = tie_scene()
    sc, osc = _programmatic_pair(pos, nrm, uv, mat, materials, [], 1, 8)
    want = tie_winner_albedo(sc.m_BVHNodes, sc.m_PrimitivesBuffer["material"], materials)
    assert np.array_equal(want, tie_winner_albedo(osc.nodes, osc.tris["material"], materials))
    old = os.environ.get("DRT_KERNEL")
    os.environ["DRT_KERNEL"] = kernel
    try:
        r = drt.Renderer(0)
    finally:
        if old is None: os.environ.pop("DRT_KERNEL", None)
        else: os.environ["DRT_KERNEL"] = old
    cam = drt.Camera((-0.5, -0.5, 5.0)); cam.m_Forward_dir = np.float32([0, 0, -1]); cam.vfov_rad = 0.0
    ocam = oracle.default_camera(position=(-0.5, -0.5, 5.0), forward=(0, 0, -1), vfov_rad=0.0)
    r.ResizeBuffer(8, 8)
    # debug albedo view: the winner's colour itself
    r.m_RendererSettings = drt.RendererSettings(RenderMode=1, DebugMode=0, ray_bounce_limit=0, tone_mapping=0, gamma_correction=0)
    r.RenderBatch(cam, sc, 1)
    img = r.GetRenderTargetImage()
    assert np.array_equal(img[..., :3], np.broadcast_to(want, (8, 8, 3))), (kernel, r.kernelInfo(), img[0, 0])
    ref, _, _ = oracle.render(osc, ocam, oracle.default_settings(render_mode=1, debug_mode=0, ray_bounce_limit=0, tone_mapping=0, gamma_correction=0), 8, 8, 1, 1)
    compare(img, ref, "tie, debug view, " + r.kernelInfo())
    # normal view, one bounce: the path leaves towards the sky with the winner's albedo as throughput
    r.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=1, tone_mapping=0, gamma_correction=0)
    r.resetAccumulationBuffer()
    r.RenderBatch(cam, sc, 1)
    img = r.GetRenderTargetImage()
    assert r.kernelInfo().startswith(kernel), r.kernelInfo()
    loser = 1.0 - want
    assert (img[..., :3] * loser).max() == 0 and (img[..., :3] * want).max() > 0, (kernel, img[0, 0])
    ref, _, _ = oracle.render(osc, ocam, oracle.default_settings(ray_bounce_limit=1, tone_mapping=0, gamma_correction=0), 8, 8, 1, 1)
    compare(img, ref, "tie, one bounce, " + r.kernelInfo())
