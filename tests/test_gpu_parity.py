"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bar (BASELINE.json north_star): per-pixel L-inf < 1e-4 on the RGBA32F framebuffer.  The design goal
is stricter -- bit-identical floats, because one ulp in the rejection sampler desynchronises a whole
path -- so every test also reports/limits the number of pixels that are not bit-equal.
"""
import re

import numpy as np
import pytest

import oracle
from tests.scenes import SCENES, bits, scene_path

pytestmark = pytest.mark.gpu
drt = pytest.importorskip("dustraytracer_amd")

LINF_TOL = 1e-4          # tolerance stated by BASELINE.json's north_star


def make_pair(name, leaf=20, bins=8):
    sc = drt.Scene()
    sc.loadGLTFmodel(scene_path(name))
    b = drt.BVHBuilder()
    b.m_TargetLeafPrimitivesCount, b.m_BinCount = leaf, bins
    b.buildIterative(sc)
    osc = oracle.Scene.load_glb(scene_path(name)).build_bvh(leaf, bins)
    return sc, osc


def cameras(name, **kw):
    _, pos, fwd, _ = SCENES[name]
    cam = drt.Camera(pos)
    cam.m_Forward_dir = np.array(fwd, np.float32)
    ocam = oracle.default_camera(position=pos, forward=fwd)
    for k, v in kw.items():
        setattr(cam, k, v)
        setattr(ocam, k, v)
    return cam, ocam


def settings_pair(**kw):
    names = {"enableSunlight": "enable_sunlight", "RenderMode": "render_mode", "DebugMode": "debug_mode"}
    s = drt.RendererSettings(**kw)
    o = oracle.default_settings(**{names.get(k, k): v for k, v in kw.items()})
    return s, o


def compare(img, ref, what, max_bit_mismatch=0):
    assert img.shape == ref.shape
    diff = np.abs(img.astype(np.float64) - ref.astype(np.float64))
    linf = float(diff.max())
    nbad = int((bits(img) != bits(ref)).any(axis=-1).sum())
    assert np.isfinite(img).all(), what
    assert linf < LINF_TOL, "%s: L-inf %.3e (%d px not bit-equal)" % (what, linf, nbad)
    assert nbad <= max_bit_mismatch, "%s: %d pixels differ in the last bits (L-inf %.3e)" % (what, nbad, linf)
    return linf, nbad


@pytest.fixture(scope="module")
def renderer():
    return drt.Renderer(0)


@pytest.mark.parametrize("name,W,H,frames", [("cornell_box", 256, 256, 1), ("cornell_box", 160, 90, 3),
                                             ("suzanne_plane", 160, 90, 2), ("dense_monkey", 160, 90, 2),
                                             ("room", 128, 72, 2), ("uv_texture_test", 128, 128, 2),
                                             ("bvh_split_test", 96, 64, 2), ("multi_material", 96, 64, 2),
                                             ("mc_transparency", 211, 115, 3), ("lightweight_rt", 128, 72, 2),
                                             ("cs16_dust", 160, 90, 2), ("sunshadow_test", 160, 90, 2),
                                             ("cornell_box_gltf", 160, 90, 2), ("uv_texture_gltf", 96, 96, 2)])
def test_image_matches_oracle(renderer, name, W, H, frames):
    sc, osc = make_pair(name)
    depth = 4 if (name, W) == ("cornell_box", 256) else SCENES[name][3]      # C1 = 256x256 1spp depth 4
    cam, ocam = cameras(name)
    s, o = settings_pair(ray_bounce_limit=depth)
    renderer.m_RendererSettings = s
    renderer.ResizeBuffer(W, H)
    renderer.resetAccumulationBuffer()
    renderer.RenderBatch(cam, sc, frames)
    assert renderer.getSampleCount() == 1 + frames
    img = renderer.GetRenderTargetImage()
    ref, ref_acc, _ = oracle.render(osc, ocam, o, W, H, 1, frames)
    compare(img, ref, name)
    compare(renderer.GetAccumulationBuffer(), ref_acc, name + " accum")
    assert (img[..., 3] == 1).all()


def test_single_frames_equal_one_batch(renderer):
    """Render() x n and RenderBatch(n) keep the same per-pixel sum order (Renderer.cu:116, RenderKernel.cu:29-30)."""
    sc, osc = make_pair("cornell_box")
    cam, ocam = cameras("cornell_box")
    s, o = settings_pair(ray_bounce_limit=3)
    renderer.m_RendererSettings = s
    renderer.ResizeBuffer(96, 54)
    renderer.resetAccumulationBuffer()
    for k in range(3):
        ms = renderer.Render(cam, sc)
        assert ms > 0 and renderer.getSampleCount() == 2 + k
        ref, _, _ = oracle.render(osc, ocam, o, 96, 54, 1, k + 1)
        compare(renderer.GetRenderTargetImage(), ref, "frame %d" % (k + 1))     # intermediate resolves too
    one_by_one = renderer.GetRenderTargetImage()
    renderer.resetAccumulationBuffer()
    assert renderer.getSampleCount() == 1
    renderer.RenderBatch(cam, sc, 3)
    assert np.array_equal(bits(one_by_one), bits(renderer.GetRenderTargetImage()))


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4, 5])
def test_debug_views(renderer, mode):
    """RayGen.cuh:137-161 incl. the BVH heat map (visit order and count must match exactly)."""
    for name in ("cornell_box", "suzanne_plane"):
        sc, osc = make_pair(name)
        cam, ocam = cameras(name)
        s, o = settings_pair(RenderMode=1, DebugMode=mode)
        renderer.m_RendererSettings = s
        renderer.ResizeBuffer(128, 72)
        renderer.resetAccumulationBuffer()
        renderer.RenderBatch(cam, sc, 2)
        ref, _, _ = oracle.render(osc, ocam, o, 128, 72, 1, 2)
        compare(renderer.GetRenderTargetImage(), ref, "%s debug %d" % (name, mode))


@pytest.mark.parametrize("kw", [dict(enableSunlight=1), dict(enableSunlight=1, tone_mapping=0),
                                dict(gamma_correction=0), dict(tone_mapping=0, gamma_correction=0),
                                dict(ray_bounce_limit=0), dict(ray_bounce_limit=-1), dict(sky_intensity=3.5, sky_color=(0.6, 0.7, 1.0))])
def test_settings_variants(renderer, kw):
    for name in ("room", "cornell_box", "suzanne_plane", "mc_transparency"):      # lds-scene and hbm-scene builds, with and without cut-outs
        sc, osc = make_pair(name)
        cam, ocam = cameras(name)
        s, o = settings_pair(**kw)
        renderer.m_RendererSettings = s
        renderer.ResizeBuffer(112, 64)
        renderer.resetAccumulationBuffer()
        renderer.RenderBatch(cam, sc, 2)
        ref, _, _ = oracle.render(osc, ocam, o, 112, 64, 1, 2)
        compare(renderer.GetRenderTargetImage(), ref, "%s %r" % (name, kw))


def test_camera_variants(renderer):
    sc, osc = make_pair("suzanne_plane")
    for kw in (dict(defocus_angle=1.2, focus_dist=4.0), dict(vfov_rad=0.6, exposure=2.5), dict(focus_dist=0.5)):
        cam, ocam = cameras("suzanne_plane", **kw)
        s, o = settings_pair()
        renderer.m_RendererSettings = s
        renderer.ResizeBuffer(120, 68)
        renderer.resetAccumulationBuffer()
        renderer.RenderBatch(cam, sc, 2)
        ref, _, _ = oracle.render(osc, ocam, o, 120, 68, 1, 2)
        compare(renderer.GetRenderTargetImage(), ref, "camera %r" % kw)


def test_alpha_cutout_changes_the_image(renderer):
    """UVtextureTest.glb holds the only RGBA texture: AnyHit must reject hits with alpha < 1 (AnyHit.cuh:8-28)."""
    sc, osc = make_pair("uv_texture_test")
    cam, ocam = cameras("uv_texture_test")
    s, o = settings_pair(ray_bounce_limit=3)
    _, _, cnt = oracle.render(osc, ocam, o, 128, 128, 1, 1, want_counters=True)
    assert cnt.anyhit_alpha > 0, "pose does not exercise the alpha path"
    for name, W, H in (("uv_texture_test", 128, 128), ("mc_transparency", 281, 153)):
        sc, osc = make_pair(name)
        cam, ocam = cameras(name)
        s, o = settings_pair(ray_bounce_limit=SCENES[name][3])
        renderer.m_RendererSettings = s
        renderer.ResizeBuffer(W, H)
        renderer.resetAccumulationBuffer()
        renderer.RenderBatch(cam, sc, 3)
        assert "lean+alpha" in renderer.kernelInfo()            # default settings + an RGBA texture in the scene
        ref, _, cnt = oracle.render(osc, ocam, o, W, H, 1, 3, want_counters=True)
        assert cnt.anyhit_alpha > 0
        compare(renderer.GetRenderTargetImage(), ref, "%s alpha cut-outs, lean kernel" % name)


def test_alpha_cutout_scene_with_sun_shadows(renderer):
    """mcTransparencyTest: RGBA leaves.  Closest-hit AND shadow traversals must skip texels with alpha < 1
    (AnyHit.cuh:8-28 from BVHTraversal.cuh:52 and :112); also the general (run-time settings) kernel variant."""
    sc, osc = make_pair("mc_transparency")
    cam, ocam = cameras("mc_transparency")
    for kw in (dict(enableSunlight=1, ray_bounce_limit=3), dict(RenderMode=1, DebugMode=3), dict(RenderMode=1, DebugMode=4)):
        s, o = settings_pair(**kw)
        renderer.m_RendererSettings = s
        renderer.ResizeBuffer(168, 92)
        renderer.resetAccumulationBuffer()
        renderer.RenderBatch(cam, sc, 2)
        assert ("lean+alpha+sun" if "enableSunlight" in kw else "general") in renderer.kernelInfo()
        ref, _, cnt = oracle.render(osc, ocam, o, 168, 92, 1, 2, want_counters=True)
        assert cnt.anyhit_alpha > 0
        compare(renderer.GetRenderTargetImage(), ref, "mc_transparency %r" % kw)


def test_other_leaf_sizes(renderer):
    for leaf, bins in ((4, 8), (8, 4), (64, 8)):
        sc, osc = make_pair("suzanne_plane", leaf, bins)
        cam, ocam = cameras("suzanne_plane")
        s, o = settings_pair()
        renderer.m_RendererSettings = s
        renderer.ResizeBuffer(96, 54)
        renderer.resetAccumulationBuffer()
        renderer.RenderBatch(cam, sc, 1)
        ref, _, _ = oracle.render(osc, ocam, o, 96, 54, 1, 1)
        compare(renderer.GetRenderTargetImage(), ref, "leaf %d bins %d" % (leaf, bins))


def test_renderer_state_machine(renderer):
    """Renderer.cu:29-78,82,132-136: resize idempotence, reset, max_samples no-op."""
    sc, _ = make_pair("room")
    cam, _ = cameras("room")
    renderer.m_RendererSettings = drt.RendererSettings(max_samples=4)
    renderer.ResizeBuffer(64, 32)
    renderer.resetAccumulationBuffer()
    renderer.Render(cam, sc)
    renderer.ResizeBuffer(64, 32)                       # same size: nothing happens
    assert renderer.getSampleCount() == 2
    renderer.RenderBatch(cam, sc, 10)                   # clamped: frame index stops AT max_samples
    assert renderer.getSampleCount() == 4
    before = renderer.GetRenderTargetImage()
    assert renderer.Render(cam, sc) == 0.0              # no-op once m_FrameIndex == max_samples
    assert renderer.getSampleCount() == 4
    assert np.array_equal(before, renderer.GetRenderTargetImage())
    renderer.ResizeBuffer(48, 32)                       # new size: realloc + reset
    assert renderer.getSampleCount() == 1 and renderer.getBufferWidth() == 48
    assert not renderer.GetAccumulationBuffer().any()


@pytest.mark.parametrize("scene,kw", [("cornell_box", dict(ray_bounce_limit=8, enableSunlight=1)), ("cornell_box", dict(ray_bounce_limit=8)),
                                      ("room", dict(ray_bounce_limit=6)), ("suzanne_plane", dict(ray_bounce_limit=3, enableSunlight=1)),
                                      ("mc_transparency", dict(ray_bounce_limit=4, enableSunlight=1)), ("uv_texture_test", dict(ray_bounce_limit=3))])
def test_counters_match_oracle(renderer, scene, kw):
    """drt_counters are counted by the tracing kernel that is measured -- path_pool's statistics build (FLAGS & 1), in every
    variant: scene in LDS / read from global memory, sunlight, alpha cut-outs -- and equal the oracle's counts exactly: SURVEY 8(d)'s
    algorithmic bytes are a function of them.  The general wave_queue kernel (debug views, the material model) counts the same."""
    sc, osc = make_pair(scene)
    cam, ocam = cameras(scene)
    s, o = settings_pair(**kw)
    ref, _, cnt = oracle.render(osc, ocam, o, 96, 54, 1, 2, want_counters=True)
    want = cnt.as_dict()
    names = ("samples", "rays", "node_visits", "inner_visits", "tri_tests", "hits_textured", "hits_flat", "shadow_rays", "inner_visits_shadow", "tri_tests_shadow")
    for kernel, r in (("path_pool", renderer), ("wave_queue", _renderer_with_env({"DRT_KERNEL": "wave_queue"}))):
        r.m_RendererSettings = s
        r.ResizeBuffer(96, 54)
        r.resetAccumulationBuffer()
        r.setCounting(True)
        try:
            r.RenderBatch(cam, sc, 2)
            got = r.getCounters().as_dict()
            assert kernel in r.kernelInfo()
            image = r.GetRenderTargetImage()
        finally:
            r.setCounting(False)
        compare(image, ref, "%s %r, counting build of %s" % (scene, kw, kernel))
        for k in names:
            assert got[k] == want[k], (kernel, k, got[k], want[k])
        if kernel == "path_pool":
            assert got["sampler_tries"] == want["sphere_iters_traced"]       # (the reference's draw after the last bounce is never used: the kernel skips it)


@pytest.mark.parametrize("stripe_rows,world", [(8, 2), (8, 8), (16, 3), (5, 4)])
def test_sharded_render_reassembles_bit_exact(stripe_rows, world):
    """Each rank renders its stripes; assembling them equals the single-device image bit for bit."""
    torch = pytest.importorskip("torch")
    W, H = 120, 70
    sc, _ = make_pair("cornell_box")
    cam, _ = cameras("cornell_box")
    full = drt.Renderer(0)
    full.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=4)
    full.ResizeBuffer(W, H)
    full.RenderBatch(cam, sc, 2)
    want = full.GetRenderTargetImage()
    padded = max(drt.shard_rows(H, stripe_rows, r, world) for r in range(world))
    gathered = torch.zeros((world, padded, W, 4), dtype=torch.float32, device="cuda:0")
    for rank in range(world):
        r = drt.Renderer(0)
        r.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=4)
        r.setShard(stripe_rows, rank, world)
        r.ResizeBuffer(W, H)
        assert r.getLocalRows() == drt.shard_rows(H, stripe_rows, rank, world)
        r.RenderBatch(cam, sc, 2)
        local = r.GetRenderTargetImage()
        gathered[rank, : local.shape[0]] = torch.from_numpy(local).cuda()
    image = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
    drt.assemble_shards(gathered.data_ptr(), image.data_ptr(), W, H, stripe_rows, world, padded,
                        torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(bits(image.cpu().numpy()), bits(want))


def test_external_buffers_and_stream():
    torch = pytest.importorskip("torch")
    W, H = 64, 40
    sc, osc = make_pair("room")
    cam, ocam = cameras("room")
    r = drt.Renderer(0)
    s, o = settings_pair()
    r.m_RendererSettings = s
    r.ResizeBuffer(W, H)
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    rgba = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.Stream()
    r.bindBuffers(accum.data_ptr(), rgba.data_ptr())
    r.setStream(stream.cuda_stream)
    r.RenderBatch(cam, sc, 2)
    stream.synchronize()
    ref, _, _ = oracle.render(osc, ocam, o, W, H, 1, 2)
    compare(rgba.cpu().numpy(), ref, "external buffers")


def _renderer_with_env(env):
    import os
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return drt.Renderer(0)                  # the knobs are read when the renderer is created
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _render_with_env(env, name, W, H, frames, depth, **settings):
    r = _renderer_with_env(env)
    sc, osc = make_pair(name)
    cam, ocam = cameras(name)
    s, o = settings_pair(ray_bounce_limit=depth, **settings)
    r.m_RendererSettings = s
    r.ResizeBuffer(W, H)
    r.RenderBatch(cam, sc, frames)
    ref, ref_acc, _ = oracle.render(osc, ocam, o, W, H, 1, frames)
    return r, ref, ref_acc


def test_batch_split_over_several_launches_keeps_sum_order():
    """A tiny sample-buffer budget forces one launch per frame; the per-pixel sum order must not change."""
    r, ref, ref_acc = _render_with_env({"DRT_SAMPLE_MB": "1"}, "cornell_box", 320, 200, 5, 4)
    compare(r.GetRenderTargetImage(), ref, "split batch")
    compare(r.GetAccumulationBuffer(), ref_acc, "split batch accum")


@pytest.mark.parametrize("votes", [(1, 1, 1, 1), (64, 64, 64, 64), (3, 60, 2, 50)])
def test_voting_thresholds_do_not_change_the_image(votes):
    """Scheduling is free (RNG state is a pure function of pixel, frame and draw count): any thresholds, same bits."""
    env = dict(zip(("DRT_VOTE_N", "DRT_VOTE_S", "DRT_VOTE_R", "DRT_VOTE_P"), map(str, votes)))
    r, ref, _ = _render_with_env(env, "room", 96, 54, 2, 6)
    compare(r.GetRenderTargetImage(), ref, "votes %r" % (votes,))


def test_exact_rcp_exhaustive():
    """device_math.hpp exact_rcp (used by the triangle test instead of the compiler's 1.0f/x expansion) must equal
    IEEE division for EVERY float: all 2^32 bit patterns are compared on the device."""
    bad, fast = drt.debug_check_rcp(0)
    assert bad == 0
    assert fast > 3_000_000_000          # the fast path really covers 2^-100 <= |x| <= 2^100 (2 * 200 * 2^23 patterns)


def test_exact_sqrt_exhaustive():
    """device_math.hpp exact_sqrt (normalize, gamma) equals the compiler's correctly rounded sqrtf for every float."""
    bad, fast = drt.debug_check_sqrt(0)
    assert bad == 0
    assert fast > 1_600_000_000          # 2^-100 <= x <= 2^100: 200 * 2^23 bit patterns


def test_full_size_properties(renderer):
    """BASELINE config C2 at full size (1920x1080, 8 spp, depth 8): size-independent properties, and the WHOLE frame
    (16.6 M samples) bit for bit against the oracle (a few seconds on the GPU box's host cores)."""
    W, H, spp = 1920, 1080, 8
    sc, osc = make_pair("cornell_box")
    cam, ocam = cameras("cornell_box")
    s, o = settings_pair(ray_bounce_limit=8)
    renderer.m_RendererSettings = s
    renderer.ResizeBuffer(W, H)
    renderer.resetAccumulationBuffer()
    renderer.RenderBatch(cam, sc, spp)
    img = renderer.GetRenderTargetImage()
    acc = renderer.GetAccumulationBuffer()
    assert np.isfinite(img).all() and (img[..., 3] == 1).all()
    assert img[..., :3].min() >= 0 and img[..., :3].max() < 1.25         # tonemapped + gamma'd samples
    assert np.array_equal(bits(img[..., :3]), bits(acc / np.float32(spp)))   # resolve = accum / frame index
    # idempotence: the same frames again give the same bits (no cross-lane state, no atomics)
    renderer.resetAccumulationBuffer()
    renderer.RenderBatch(cam, sc, spp)
    assert np.array_equal(bits(img), bits(renderer.GetRenderTargetImage()))
    ref, ref_acc, _ = oracle.render(osc, ocam, o, W, H, 1, spp)
    compare(img, ref, "C2 whole frame")
    assert np.array_equal(bits(acc), bits(ref_acc))


@pytest.mark.parametrize("name,W,H,spp,depth,stripe_world", [("suzanne_plane", 1920, 1080, 8, 2, 1),         # BASELINE configs[2]
                                                            ("dense_monkey", 1920, 1080, 16, 2, 1),        # configs[3]
                                                            ("cs16_dust", 1920, 1080, 2, 5, 1),            # the large closed map
                                                            ("room", 3840, 2160, 64, 16, 54)])             # configs[4]
def test_other_baseline_configs_at_full_size(renderer, name, W, H, spp, depth, stripe_world):
    """Full-size BASELINE configs: frame-wide invariants, and the frame against the oracle -- all of it (stripe_world 1)
    where the host finishes in seconds, else the rows of every stripe_world-th 8-row stripe (room: 531 M samples)."""
    sc, osc = make_pair(name)
    cam, ocam = cameras(name)
    s, o = settings_pair(ray_bounce_limit=depth, max_samples=spp + 1)
    renderer.m_RendererSettings = s
    renderer.ResizeBuffer(W, H)
    renderer.resetAccumulationBuffer()
    renderer.RenderBatch(cam, sc, spp)
    assert renderer.getSampleCount() == spp + 1
    img = renderer.GetRenderTargetImage()
    acc = renderer.GetAccumulationBuffer()
    assert np.isfinite(img).all() and (img[..., 3] == 1).all()
    assert img[..., :3].min() >= 0 and img[..., :3].max() < 1.25
    assert np.array_equal(bits(img[..., :3]), bits(acc / np.float32(spp)))
    ref, _, _ = oracle.render(osc, ocam, o, W, H, 1, spp, stripe_rows=8, rank=stripe_world // 2, world=stripe_world)
    rows = np.array([y for y in range(H) if (y // 8) % stripe_world == stripe_world // 2])
    compare(img[rows], ref[rows], "%s full size band" % name)


def test_headless_cli_example_writes_the_image(tmp_path):
    """examples/drt_render.cpp (C++ wrapper classes over the C ABI): scene.glb -> PNG, as the editor's "save png" would."""
    import os
    import subprocess
    from PIL import Image
    from tests.scenes import ROOT
    exe = tmp_path / "drt_render"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "drt_render.cpp"),
                    "-L" + os.path.join(ROOT, "dustraytracer_amd"), "-ldrt_hip", "-Wl,-rpath," + os.path.join(ROOT, "dustraytracer_amd"),
                    "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    out = tmp_path / "cornell.png"
    name = "cornell_box"
    _, pos, fwd, _ = SCENES[name]
    args = [str(exe), scene_path(name), str(out), "96", "64", "2", "4"] + [str(v) for v in pos] + [str(v) for v in fwd]
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    png = np.asarray(Image.open(out))
    assert png.shape == (64, 96, 4)
    osc = oracle.Scene.load_glb(scene_path(name)).build_bvh(20, 8)
    ref, _, _ = oracle.render(osc, oracle.default_camera(position=pos, forward=fwd), oracle.default_settings(ray_bounce_limit=4), 96, 64, 1, 2)
    want = (np.clip(ref[::-1], 0, 1) * np.float32(255) + np.float32(0.5)).astype(np.uint8)
    assert np.array_equal(png, want)


def test_kernel_span_is_within_the_stream_event_time(renderer):
    """drt_renderer_kernel_span: the tracing kernel's own first-wave-in .. last-wave-out time; for a launch that has the GPU
    to itself it must lie inside the HIP-event bracket around the launch (which also holds the resolve kernel)."""
    sc, _ = make_pair("cornell_box")
    cam, _ = cameras("cornell_box")
    renderer.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=8)
    renderer.ResizeBuffer(640, 360)
    for _ in range(3):
        renderer.resetAccumulationBuffer()
        ms = renderer.RenderBatch(cam, sc, 4)
        span = renderer.kernelSpanMs()
    assert 0.0 < span <= ms * 1.02 + 0.01, (span, ms)
    assert span > 0.5 * ms, (span, ms)


def _programmatic_pair(pos, nrm, uv, mat, materials, textures, leaf, bins):
    """The same de-indexed geometry given to the product (drt_scene_set_geometry / add_material / add_texture) and to the oracle."""
    sc = drt.Scene()
    for tex in textures:
        sc.addTexture(tex)
    for alb, tex in materials:
        sc.addMaterial(alb, tex)
    sc.setGeometry(pos, nrm, uv, mat)
    b = drt.BVHBuilder()
    b.m_TargetLeafPrimitivesCount, b.m_BinCount = leaf, bins
    b.buildIterative(sc)
    n = len(mat)
    tris = np.zeros(n, oracle.TRI_DTYPE)
    a = [np.ascontiguousarray(x, np.float32) for x in (pos.reshape(-1, 3), nrm.reshape(-1, 3), uv.reshape(-1, 2))]
    m = np.ascontiguousarray(mat, np.int32)
    oracle.lib().o_build_triangles(a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, m.ctypes.data, n, tris.ctypes.data)
    osc = oracle.Scene(tris, materials, textures).build_bvh(leaf, bins)
    return sc, osc


def _render_both(renderer, sc, osc, pos, fwd, W, H, frames, **settings):
    cam = drt.Camera(pos)
    cam.m_Forward_dir = np.array(fwd, np.float32)
    s, o = settings_pair(**settings)
    renderer.m_RendererSettings = s
    renderer.ResizeBuffer(W, H)
    renderer.resetAccumulationBuffer()
    renderer.RenderBatch(cam, sc, frames)
    ref, _, _ = oracle.render(osc, oracle.default_camera(position=pos, forward=fwd), o, W, H, 1, frames)
    return renderer.GetRenderTargetImage(), ref


def test_edge_case_scenes_and_frame_sizes(renderer):
    """Inputs at the edges: one triangle, a 1x1 and a 7x3 frame, a deep tree (stack > 16 levels, scene not in LDS), a long
    path (32 bounces inside a closed box), RGB and RGBA textures given as arrays."""
    rng = np.random.default_rng(5)
    up = np.tile(np.float32([0, 0, 1]), (3, 1))
    # (1) a single triangle, tiny frames
    pos = np.float32([[[-1, -1, 0], [1, -1, 0], [0, 1, 0]]])
    sc, osc = _programmatic_pair(pos, up[None], np.zeros((1, 3, 2), np.float32), [0], [((0.8, 0.3, 0.2), -1)], [], 20, 8)
    assert len(sc.m_BVHNodes) == 1
    for W, H in ((1, 1), (7, 3), (64, 1), (1, 64)):
        img, ref = _render_both(renderer, sc, osc, (0.1, 0.0, 2.0), (0, 0, -1), W, H, 3, ray_bounce_limit=3)
        compare(img, ref, "single triangle %dx%d" % (W, H))
    # (2) triangle soup with small leaves: deep tree
    n = 20000
    c = rng.uniform(-4, 4, (n, 1, 3)).astype(np.float32)
    pos = c + rng.uniform(-0.25, 0.25, (n, 3, 3)).astype(np.float32)
    nrm = rng.normal(size=(n, 3, 3)).astype(np.float32)
    uv = rng.uniform(-2, 3, (n, 3, 2)).astype(np.float32)
    tex_rgb = rng.integers(0, 256, (19, 23, 3), dtype=np.uint8)
    tex_rgba = rng.integers(0, 256, (8, 8, 4), dtype=np.uint8)
    tex_rgba[..., 3] = np.where(rng.uniform(size=(8, 8)) < 0.5, 255, 40)
    materials = [((0.9, 0.9, 0.9), -1), ((1.0, 1.0, 1.0), 0), ((0.7, 0.8, 0.9), 1)]
    mat = rng.integers(0, 3, n).astype(np.int32)
    sc, osc = _programmatic_pair(pos, nrm, uv, mat, materials, [tex_rgb, tex_rgba], 3, 8)
    assert sc.bvh_depth > 16
    img, ref = _render_both(renderer, sc, osc, (0.0, 0.5, 9.0), (0, -0.05, -1), 96, 54, 2, ray_bounce_limit=6)
    compare(img, ref, "soup depth %d" % sc.bvh_depth)
    assert renderer.kernelInfo().startswith("path_pool<lean+alpha,hbm-scene>") and re.search(r"stack=%d[ (]" % (sc.bvh_depth - 1), renderer.kernelInfo()), renderer.kernelInfo()
    img, ref = _render_both(renderer, sc, osc, (0.0, 0.5, 9.0), (0, -0.05, -1), 96, 54, 2, ray_bounce_limit=6, enableSunlight=1)
    compare(img, ref, "soup with sun shadows through cut-outs")
    # (2b) a small scene under a degenerate tree: 62 triangles whose centroids double in x peel off one per level (leaf size 1, two
    #      bins) -- the scene fits LDS many times over, the traversal stacks of a tree this deep take most of it
    n = 62
    cx = (2.0 ** np.arange(n)).astype(np.float32)
    pos = np.zeros((n, 3, 3), np.float32)
    pos[:, :, 0] = cx[:, None]
    pos += (np.float32([[0, -1, -1], [0, 1, -1], [0, 0, 1]]) * 0.4)[None] * cx[:, None, None]
    nrm = np.tile(np.float32([-1, 0, 0]), (n, 3, 1))
    sc, osc = _programmatic_pair(pos, nrm, np.zeros((n, 3, 2), np.float32), np.zeros(n, np.int32), [((0.7, 0.6, 0.5), -1)], [], 1, 2)
    assert 32 < sc.bvh_depth <= 64
    for kw in ({}, {"enableSunlight": 1}):
        img, ref = _render_both(renderer, sc, osc, (-3.0, 0.1, 0.2), (1.0, 0.02, -0.03), 64, 36, 2, ray_bounce_limit=3, **kw)
        compare(img, ref, "degenerate %d-level tree %r (%s)" % (sc.bvh_depth, kw, renderer.kernelInfo()))
        # (path_pool keeps the near child of a visit in registers: one stack slot per level below the root)
        assert re.search(r"stack=%d[ (]" % (sc.bvh_depth - 1 if renderer.kernelInfo().startswith("path_pool") else sc.bvh_depth), renderer.kernelInfo())
        if renderer.kernelInfo().startswith("path_pool"):
            assert " in LDS)" in renderer.kernelInfo(), renderer.kernelInfo()      # (a tree this deep: most stack levels live in HBM)
    # (3) long paths: the closed cornell box with 32 bounces
    sc, osc = make_pair("cornell_box")
    cam, ocam = cameras("cornell_box")
    s, o = settings_pair(ray_bounce_limit=32)
    renderer.m_RendererSettings = s
    renderer.ResizeBuffer(80, 45)
    renderer.resetAccumulationBuffer()
    renderer.RenderBatch(cam, sc, 2)
    ref, _, _ = oracle.render(osc, ocam, o, 80, 45, 1, 2)
    compare(renderer.GetRenderTargetImage(), ref, "32 bounces")


def test_randomised_cases_match_oracle():
    """tools/fuzz_parity.py, a short run: random scene / camera / lens / settings / frame size, all kernel variants."""
    import os
    import subprocess
    import sys
    from tests.scenes import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "250", "31"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatching 0" in r.stdout


def test_bench_line_has_the_contracted_shape():
    """bench.py prints ONE JSON line with the keys the driver reads, the roofline and cpu_baseline objects included."""
    import json
    import os
    import subprocess
    import sys
    from tests.scenes import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--cpu-seconds", "1"],
                       capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key, kind in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                      ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], kind), key
    assert d["metric"].startswith("Msamples/sec at 1920x1080, 8spp, cornell_box") and d["unit"] == "Msamples/s"
    assert (d["n_gpus"], d["steps"], d["warmup"], d["higher_is_better"], d["scaling"], d["vs_baseline"], d["dtype"]) == (1, 6, 2, True, "strong", None, "f32")
    assert d["config"]["workload"] == "cornell_box_1080p_8spp_d8" and "model" not in d["config"]
    assert abs(d["value"] - 1920 * 1080 * 8 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]
    roof = d["roofline"]
    # the bound that binds: vector-ALU issue (counters from profiles/, kernel time from this run); the HBM line of SURVEY 8(d) rides along
    assert roof["bound"] == "valu_issue" and roof["peak"] == 1228.8 and roof["kernel_ms"] > 0 and roof["launches_per_step"] == 1
    assert roof["kernel_ms"] <= d["ms_per_step"] * 1.5                 # the kernel alone, not a span stretched by other frames in flight
    if roof["achieved"] is not None:                                   # profiles/r02_roofline_<workload>.json present
        assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and 0 < roof["frac"] <= 1.0
        assert abs(roof["achieved"] - (roof["valu_wave_instructions_per_launch"] * 1.0) / (roof["kernel_ms"] * 1e-3) / 1e9) < 0.25 * roof["achieved"]
        assert 0.5 < roof["lane_utilisation"] <= 1.0
    hbm = roof["hbm"]
    assert hbm["peak"] == 8000.0 and hbm["unit"] == "GB/s" and hbm["served_from"] == "lds"
    assert abs(hbm["achieved"] - hbm["algorithmic_bytes_per_launch"] / (roof["kernel_ms"] * 1e-3) / 1e9) < 0.01 * hbm["achieved"]
    assert abs(hbm["frac"] - hbm["achieved"] / hbm["peak"]) < 1e-3
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["unit"] == "Msamples/s" and cpu["value"] > 0 and cpu["cores"] >= 1 and "samples" in cpu["sample"]


def test_bench_two_rank_rehearsal_assembles_the_single_device_image():
    """The multi-GPU code path of bench.py on the one-GPU box: two ranks (torch.distributed.run), both on device 0, stripes
    gathered through gloo (RCCL refuses two ranks on one device) and assembled on rank 0 -- the result must be the
    single-device image, bit for bit."""
    import json
    import os
    import subprocess
    import sys
    from tests.scenes import ROOT
    env = dict(os.environ, DRT_BENCH_REHEARSAL="1", DRT_BENCH_CHECK="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "sharded image == single-device image: True" in r.stderr
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "stripes8x2" and "REHEARSAL" in d["data"]


@pytest.mark.gpu
def test_kernel_packaging_is_measured_not_guessed():
    """wave_queue has several legal packagings of a launch (workgroup size, stack entry bytes, triangles per T step; DESIGN.md 4 /
    5.2).  They all compute the same image, so the renderer times each on its first big launches and keeps the fastest (path_pool's
    shape follows from what fits LDS).  Checked: the image is the oracle's whatever is being tried, every candidate gets its trials,
    and the one kept is the best measured (within 2 %)."""
    expect = {"cornell_box": "path_pool<lean,lds-scene> stack=2 wg/CU=2 ",             # small LDS scene, lean paths: the path pool, two pools per CU
              "room": "path_pool<lean,lds-scene> stack=7(1 in LDS) wg/CU=2 threads=768 paths=1024 ",   # 17.6 KB LDS scene, 8-level tree: two pools per CU, all but one stack level in HBM
              # trees read from global memory: the narrow kernels (80 VGPRs, two triangles per T step), two pools of 12 waves per CU, 6-byte stack entries of which three levels in LDS
              "cs16_dust": "path_pool<lean,hbm-scene> stack=15(3 in LDS) wg/CU=2 threads=768 paths=960 ",
              "suzanne_plane": "path_pool<lean,hbm-scene> stack=9(3 in LDS) wg/CU=2 threads=768 paths=960 "}
    r = drt.Renderer(0)
    r_wq = _renderer_with_env({"DRT_KERNEL": "wave_queue"})
    W, H = 160, 90
    r.ResizeBuffer(W, H)
    for name, marker in expect.items():
        sc, osc = make_pair(name)
        cam, ocam = cameras(name)
        s, o = settings_pair(ray_bounce_limit=3, max_samples=100)
        r.m_RendererSettings = s
        r.resetAccumulationBuffer()
        r.RenderBatch(cam, sc, 2)
        assert marker in r.kernelInfo(), (name, r.kernelInfo())
        ref, _, _ = oracle.render(osc, ocam, o, W, H, 1, 2)
        compare(r.GetRenderTargetImage(), ref, name)
    assert all(p["chosen"] == -1 or len(p["candidates"]) == 1 for p in r.waveQueuePlans())      # 160x90 launches are too small to time
    # big launches of one scene on wave_queue: every candidate is tried (bit-exact each time), then one is kept
    r = r_wq
    sc, osc = make_pair("cs16_dust")
    cam, ocam = cameras("cs16_dust")
    s, o = settings_pair(ray_bounce_limit=2, max_samples=1000)
    r.m_RendererSettings = s
    W, H = 640, 360
    r.ResizeBuffer(W, H)
    ref, _, _ = oracle.render(osc, ocam, o, W, H, 1, 2)
    seen = set()
    for _ in range(10):
        r.resetAccumulationBuffer()
        r.RenderBatch(cam, sc, 2)
        seen.add(r.kernelInfo())
        compare(r.GetRenderTargetImage(), ref, "cs16_dust " + r.kernelInfo())
    plans = [p for p in r.waveQueuePlans() if len(p["candidates"]) > 1 and all(c["trials"] >= 2 for c in p["candidates"])]
    assert plans, r.waveQueuePlans()
    assert len(seen) >= 2, seen                                       # more than one packaging really ran
    for p in plans:
        best = min(c["ns_per_sample"] for c in p["candidates"])
        assert p["chosen"] >= 0 and p["candidates"][p["chosen"]]["ns_per_sample"] <= 1.02 * best, p


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["path_pool", "wave_queue"])
@pytest.mark.parametrize("name,W,H,frames,depth", [("cornell_box", 160, 90, 3, 8), ("room", 128, 72, 2, 16), ("bvh_split_test", 96, 64, 2, 4),
                                                    ("cornell_box", 61, 37, 2, 0), ("room", 64, 64, 1, 32),
                                                    ("suzanne_plane", 160, 90, 2, 3), ("cs16_dust", 128, 72, 2, 5), ("dense_monkey", 96, 54, 2, 4),
                                                    ("mc_transparency", 128, 72, 2, 5)])
def test_both_tracing_kernels_match_the_oracle_on_lds_scenes(kernel, name, W, H, frames, depth):
    """Every scene is traced by path_pool (kernel_path_pool.hip: path state parked in LDS, phase-homogeneous batches; the
    traversal data in LDS too where it fits, else read from global memory) unless DRT_KERNEL=wave_queue; both are bit-exact,
    partial tiles and long paths included."""
    r, ref, ref_acc = _render_with_env({"DRT_KERNEL": kernel}, name, W, H, frames, depth)
    assert r.kernelInfo().startswith(kernel), r.kernelInfo()
    compare(r.GetRenderTargetImage(), ref, "%s %s" % (kernel, name))
    compare(r.GetAccumulationBuffer(), ref_acc, "%s %s accum" % (kernel, name))


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"DRT_POOL_THREADS": "64", "DRT_POOL_PATHS": "64"}, {"DRT_POOL_THREADS": "1024", "DRT_POOL_PATHS": "256"},
                                 {"DRT_POOL_MIN_FILL": "1", "DRT_POOL_PATIENCE": "0", "DRT_POOL_N_LOOP": "1"},
                                 {"DRT_POOL_MIN_FILL": "64", "DRT_POOL_PATIENCE": "100", "DRT_POOL_DIR_TRIES": "1", "DRT_POOL_N_LOOP": "64", "DRT_POOL_N_MIN": "1"},
                                 {"DRT_POOL_DIR_TRIES": "2000", "DRT_POOL_COLD_KB": "0"}, {"DRT_POOL_PATHS": "4032", "DRT_POOL_COLD_KB": "64"}])
def test_path_pool_scheduling_knobs_do_not_change_the_image(env):
    """Pool size, workgroup size, batch thresholds, candidates per batch: scheduling only (the RNG state is a pure function of
    pixel, frame and draw count), so every setting gives the oracle's bits."""
    r, ref, _ = _render_with_env(dict(env, DRT_KERNEL="path_pool"), "room", 96, 54, 2, 6)
    assert r.kernelInfo().startswith("path_pool"), r.kernelInfo()
    compare(r.GetRenderTargetImage(), ref, "pool knobs %r" % (env,))


@pytest.mark.gpu
@pytest.mark.parametrize("levels", ["1", "2", "4"])
@pytest.mark.parametrize("name,W,H,frames,depth,sun", [("room", 96, 54, 2, 8, 0), ("room", 96, 54, 2, 6, 1), ("cs16_dust", 96, 54, 2, 4, 0), ("cs16_dust", 96, 54, 2, 4, 1),
                                                        ("dense_monkey", 96, 54, 2, 3, 0), ("mc_transparency", 96, 54, 2, 4, 1), ("suzanne_plane", 96, 54, 2, 2, 0)])
def test_stack_levels_in_hbm_do_not_change_the_image(levels, name, W, H, frames, depth, sun):
    """Deep trees: path_pool keeps only the bottom levels of the traversal stacks in LDS and the rest in HBM (chosen by the launcher;
    forced here to 1, 2 and 4 levels so that nearly every push and pop of these scenes goes through the HBM part): where a stack
    entry lives is not arithmetic -- the oracle's bits, closest-hit and shadow traversals, scene in LDS and read from global memory."""
    r, ref, ref_acc = _render_with_env({"DRT_KERNEL": "path_pool", "DRT_POOL_STACK_LDS": levels}, name, W, H, frames, depth, enableSunlight=sun)
    assert r.kernelInfo().startswith("path_pool") and "(%s in LDS)" % levels in r.kernelInfo(), r.kernelInfo()
    compare(r.GetRenderTargetImage(), ref, "%s, %s stack levels in LDS, sun=%d" % (name, levels, sun))
    compare(r.GetAccumulationBuffer(), ref_acc, "%s accum" % name)


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["path_pool", "wave_queue"])
@pytest.mark.parametrize("name,W,H,frames,depth,sun", [("cornell_box", 128, 72, 2, 5, 1), ("room", 96, 54, 2, 8, 1), ("sunshadow_test", 128, 72, 2, 3, 1),
                                                        ("uv_texture_test", 128, 72, 2, 5, 0), ("uv_texture_test", 96, 54, 2, 4, 1),
                                                        ("cornell_box", 64, 36, 1, 0, 1),
                                                        ("cs16_dust", 96, 54, 2, 4, 1), ("mc_transparency", 96, 54, 2, 4, 1), ("suzanne_plane", 96, 54, 2, 2, 1)])
def test_sunlight_and_alpha_cutouts_on_both_tracing_kernels(kernel, name, W, H, frames, depth, sun):
    """The sun's shadow ray (RayGen.cuh:124-128: RayTest, any accepted hit occludes) and the alpha test of AnyHit.cuh:8-28 in
    path_pool (shadow traversals ride the same N / T queues, S = the rest of the shading) and in wave_queue: the oracle's bits."""
    r, ref, ref_acc = _render_with_env({"DRT_KERNEL": kernel}, name, W, H, frames, depth, enableSunlight=sun)
    assert r.kernelInfo().startswith(kernel), r.kernelInfo()
    want = "lean" + ("+alpha" if name in ("uv_texture_test", "mc_transparency") else "") + ("+sun" if sun else "")
    assert "<%s," % want in r.kernelInfo(), r.kernelInfo()
    compare(r.GetRenderTargetImage(), ref, "%s %s sun=%d" % (kernel, name, sun))
    compare(r.GetAccumulationBuffer(), ref_acc, "%s %s accum" % (kernel, name))


@pytest.mark.gpu
def test_path_pool_falls_back_where_it_does_not_apply():
    """Debug views (and the counting build, the material model: tests of their own) stay on wave_queue."""
    for name, kw in (("cornell_box", dict(RenderMode=1, DebugMode=1)), ("suzanne_plane", dict(RenderMode=1, DebugMode=4)), ("cs16_dust", dict(RenderMode=1, DebugMode=0))):
        sc, osc = make_pair(name)
        cam, ocam = cameras(name)
        s, o = settings_pair(ray_bounce_limit=3, **kw)
        r = drt.Renderer(0)
        r.m_RendererSettings = s
        r.ResizeBuffer(96, 54)
        r.RenderBatch(cam, sc, 2)
        assert r.kernelInfo().startswith("wave_queue"), (name, kw, r.kernelInfo())
        ref, _, _ = oracle.render(osc, ocam, o, 96, 54, 1, 2)
        compare(r.GetRenderTargetImage(), ref, "%s %r" % (name, kw))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell_box", "suzanne_plane"])
def test_recursive_build_order_renders_the_same_image(renderer, name):
    """BVHBuilder::build numbers the nodes differently from buildIterative (BVHBuilder.cu:100-173 vs :11-92); the tree is the same
    tree, so every traversal visits the same boxes and triangles in the same order: same image, bit for bit."""
    sc = drt.Scene(); sc.loadGLTFmodel(scene_path(name))
    b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8
    b.build(sc)
    osc = oracle.Scene.load_glb(scene_path(name)).build_bvh(20, 8, recursive=True)
    cam, ocam = cameras(name)
    s, o = settings_pair(ray_bounce_limit=4)
    renderer.m_RendererSettings = s
    renderer.ResizeBuffer(128, 72)
    renderer.resetAccumulationBuffer()
    renderer.RenderBatch(cam, sc, 2)
    ref, _, _ = oracle.render(osc, ocam, o, 128, 72, 1, 2)
    compare(renderer.GetRenderTargetImage(), ref, "recursive build order, oracle with the same order")
    ref2, _, _ = oracle.render(oracle.Scene.load_glb(scene_path(name)).build_bvh(20, 8), ocam, o, 128, 72, 1, 2)
    assert np.array_equal(bits(ref), bits(ref2))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["suzanne_plane_gltf", "scene_hier_test", "lightweight_rt"])
def test_strict_loaded_scene_renders_like_the_strict_oracle_scene(renderer, name):
    """DRT_LOAD_STRICT end to end: the file read as the glTF specification says (node transforms, hierarchy, accessor rules) by the
    product and by the oracle's independent strict restatement, built, rendered: same image bit for bit -- and, for the scene whose
    node translations the reference drops (suzanne_plane.gltf), not the image of the default loader."""
    path = scene_path(name)
    sc = drt.Scene(); sc.loadGLTFmodel(path, strict=True)
    b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8; b.buildIterative(sc)
    osc = oracle.Scene.load_glb(path, strict=True).build_bvh(20, 8)
    cam, ocam = cameras(name)
    s, o = settings_pair(ray_bounce_limit=3)
    renderer.m_RendererSettings = s
    renderer.ResizeBuffer(128, 72)
    renderer.resetAccumulationBuffer()
    renderer.RenderBatch(cam, sc, 2)
    img = renderer.GetRenderTargetImage()
    ref, _, _ = oracle.render(osc, ocam, o, 128, 72, 1, 2)
    compare(img, ref, "strict " + name)
    if name == "suzanne_plane_gltf":
        ref_default, _, _ = oracle.render(oracle.Scene.load_glb(path).build_bvh(20, 8), ocam, o, 128, 72, 1, 2)
        assert not np.array_equal(bits(ref), bits(ref_default))
