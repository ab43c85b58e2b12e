"""the material features the reference loads and never reads (Scene.cu:71-75: emissiveFactor, roughnessFactor,
metallicFactor) behind an opt-in switch (include/drt.h drt_material_model; oracle: o_scene.ext_*).  Off -- the default -- is the
reference's image bit for bit; on, the product's general kernel must agree with the oracle's restatement of the same rule.
The rule itself is ours (the reference has none): documented in drt.h, hand-checked here on cases small enough to derive."""
import numpy as np
import pytest

import oracle
from tests.scenes import SCENES, bits, scene_path

drt = pytest.importorskip("dustraytracer_amd")


def _one_triangle_scene(emissive, metallic, roughness, albedo=(0.5, 0.25, 0.125)):
    tris = np.zeros(1, oracle.TRI_DTYPE)
    pos = np.float32([[-4, -4, 0.5], [4, -4, 0.5], [-4, 4, 0.5]])
    nrm = np.tile(np.float32([0, 0, 1]), (3, 1))
    mat = np.int32([0])
    oracle.lib().o_build_triangles(pos.ctypes.data, nrm.ctypes.data, np.zeros((3, 2), np.float32).ctypes.data, mat.ctypes.data, 1, tris.ctypes.data)
    return oracle.Scene(tris, [(np.float32(albedo), -1)], [], [(np.float32(emissive), np.float32(roughness), int(metallic))]).build_bvh(20, 8)


def test_oracle_emissive_term_is_emission_times_the_throughput_so_far():
    """Hand-derived: zero field of view, straight down onto one emissive triangle, no bounce (ray_bounce_limit 0), no post-processing:
    the sample is exactly emissive * scale (throughput is still 1 at the first hit); switched off it is black."""
    osc = _one_triangle_scene((0.25, 2.0, 0.0), 0, 0.5)
    cam = oracle.default_camera(position=(-0.5, -0.5, 5.0), forward=(0, 0, -1), vfov_rad=0.0)
    st = oracle.default_settings(ray_bounce_limit=0, tone_mapping=0, gamma_correction=0)
    off, _, _ = oracle.render(osc, cam, st, 4, 4, 1, 1)
    assert not off[..., :3].any()
    osc.material_model = (1, 0, 3.0)
    on, _, _ = oracle.render(osc, cam, st, 4, 4, 1, 1)
    assert np.array_equal(on[..., :3], np.broadcast_to(np.float32([0.75, 6.0, 0.0]), (4, 4, 3)))
    # one bounce: the emission of the first hit plus the sky seen through the albedo -- more light than without emission, by exactly that term
    st1 = oracle.default_settings(ray_bounce_limit=1, tone_mapping=0, gamma_correction=0)
    on1, _, _ = oracle.render(osc, cam, st1, 4, 4, 1, 1)
    osc.material_model = (0, 0, 1.0)
    off1, _, _ = oracle.render(osc, cam, st1, 4, 4, 1, 1)
    assert np.allclose(on1[..., :3] - off1[..., :3], np.float32([0.75, 6.0, 0.0]), rtol=1e-6, atol=1e-6)


def test_oracle_mirror_lobe_reflects_about_the_normal():
    """A metallic triangle of roughness 0 seen straight down reflects the ray straight up: the sample is the zenith sky times the
    albedo, whatever the frame index (no randomness survives a zero roughness); a diffuse one depends on the frame."""
    osc = _one_triangle_scene((0, 0, 0), 1, 0.0, albedo=(1.0, 0.5, 0.25))
    cam = oracle.default_camera(position=(-0.5, -0.5, 5.0), forward=(0, 0, -1), vfov_rad=0.0)
    st = oracle.default_settings(ray_bounce_limit=1, tone_mapping=0, gamma_correction=0)
    osc.material_model = (0, 1, 1.0)
    _, a, _ = oracle.render(osc, cam, st, 4, 4, 1, 1)                    # (the accumulation buffer: the frame's samples themselves)
    _, b, _ = oracle.render(osc, cam, st, 4, 4, 5, 1)
    assert np.array_equal(bits(a), bits(b))
    # the reflected ray runs along +z: SkyModel's lerp factor 0.5 (1 + d.y) is 0.5 there -> ((1 + sky_color) / 2)^2 * intensity
    sky = ((np.float32(1.0) + np.float32(st.sky_color)) * np.float32(0.5)) ** 2 * np.float32(st.sky_intensity)
    assert np.allclose(a[0, 0, :3], sky * np.float32([1.0, 0.5, 0.25]), rtol=1e-6)
    osc.material_model = (0, 0, 1.0)
    _, c, _ = oracle.render(osc, cam, st, 4, 4, 1, 1)
    _, d, _ = oracle.render(osc, cam, st, 4, 4, 5, 1)
    assert not np.array_equal(bits(c), bits(d))
    # a mirror looked at from below its normal side still reflects (the normal is turned towards the ray, ClosestHit.cuh:17-24)
    cam2 = oracle.default_camera(position=(-0.5, -0.5, -5.0), forward=(0, 0, 1), vfov_rad=0.0)
    osc.material_model = (0, 1, 1.0)
    e, _, _ = oracle.render(osc, cam2, st, 2, 2, 1, 1)
    assert np.isfinite(e).all() and e[..., :3].max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["emissive_test", "cornell_box_gltf", "cs16_dust"])
@pytest.mark.parametrize("model", [(1, 0, 1.0), (0, 1, 1.0), (1, 1, 2.5)])
def test_material_model_matches_the_oracle_and_is_off_by_default(name, model):
    """On: the general wave_queue kernel == the oracle with the same switches, bit for bit (EmissiveTest.glb: emissive cubes and a
    mirror sphere; cornell_box.gltf: the ceiling light mesh; cs16_dust: every glTF-default material is 'metallic', roughness 1).
    Off again: the very image a renderer that never heard of the switch produces."""
    sc = drt.Scene(); sc.loadGLTFmodel(scene_path(name))
    b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8; b.buildIterative(sc)
    osc = oracle.Scene.load_glb(scene_path(name)).build_bvh(20, 8)
    _, pos, fwd, depth = SCENES[name]
    cam = drt.Camera(pos); cam.m_Forward_dir = np.array(fwd, np.float32)
    ocam = oracle.default_camera(position=pos, forward=fwd)
    W, H, frames = 128, 72, 2
    for sun in (0, 1):
        st = oracle.default_settings(ray_bounce_limit=depth, enable_sunlight=sun)
        r = drt.Renderer(0)
        r.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=depth, enableSunlight=sun)
        r.ResizeBuffer(W, H)
        r.RenderBatch(cam, sc, frames)
        plain = r.GetRenderTargetImage()
        ref_plain, _, _ = oracle.render(osc, ocam, st, W, H, 1, frames)
        assert np.array_equal(bits(plain), bits(ref_plain))
        r.setMaterialModel(*model)
        r.resetAccumulationBuffer()
        r.RenderBatch(cam, sc, frames)
        img = r.GetRenderTargetImage()
        assert "general" in r.kernelInfo(), r.kernelInfo()
        osc.material_model = model
        ref, _, _ = oracle.render(osc, ocam, st, W, H, 1, frames)
        osc.material_model = (0, 0, 1.0)
        nbad = int((bits(img) != bits(ref)).any(axis=-1).sum())
        assert nbad == 0, "%s %r sun=%d: %d pixels differ from the oracle" % (name, model, sun, nbad)
        mats = sc.m_Material
        touches = (model[0] and np.asarray(mats["emissive"]).any()) or (model[1] and np.asarray(mats["metallic"]).any())
        if touches:                                                          # (a scene without such materials looks the same)
            assert not np.array_equal(bits(img), bits(plain))                # the switch does change the image
        else:
            assert np.array_equal(bits(img), bits(plain))
        r.setMaterialModel(0, 0, 1.0)
        r.resetAccumulationBuffer()
        r.RenderBatch(cam, sc, frames)
        assert np.array_equal(bits(r.GetRenderTargetImage()), bits(plain))
