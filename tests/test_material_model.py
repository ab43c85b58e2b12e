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


def _slab_scene(ior, tint=(1.0, 1.0, 1.0)):
    """Two big parallel triangles, z = 1 and z = 0, both of one transmissive material: a slab of glass seen from above."""
    pos = np.float32([[-40, -40, 1], [40, -40, 1], [-40, 40, 1], [-40, -40, 0], [40, -40, 0], [-40, 40, 0]])
    nrm = np.float32([[0, 0, 1]] * 3 + [[0, 0, -1]] * 3)                   # outward normals: front face = entering the glass (ClosestHit.cuh:17-24)
    tris = np.zeros(2, oracle.TRI_DTYPE)
    oracle.lib().o_build_triangles(pos.ctypes.data, nrm.ctypes.data, np.zeros((6, 2), np.float32).ctypes.data, np.int32([0, 0]).ctypes.data, 2, tris.ctypes.data)
    return oracle.Scene(tris, [(np.float32(tint), -1)], [], [(np.float32((0, 0, 0)), np.float32(0.0), 0, 1, np.float32(ior))]).build_bvh(20, 8)


def test_oracle_dielectric_slab_at_normal_incidence_lets_the_ray_through_unbent():
    """Hand-derived.  Zero field of view, straight down (0, 0, -1) onto a slab of refractive index 1.5.  cos = 1, so refract() gives
    r_perp = ri * (v + n) = 0 and r_par = -sqrt(|1 - 0|) n = -n: the ray goes on straight down, through both faces (Random.cu:26-32),
    and leaves along -z with the tint applied twice -- unless Schlick's reflectance (Random.cu:34-40), r0 = ((1 - ri) / (1 + ri))^2 =
    0.04 at both faces, exceeds the face's random number.  SkyModel only looks at d.y, which is 0 along +-z, so every sample is one
    of: sky * tint^2 (transmitted twice), sky * tint (reflected at the first face), or a longer path that bounced inside (odd powers
    >= 3 of the tint going up, even ones >= 4 going down); over many frames the first outcome is seen in about 0.96 * 0.96 = 92 % of
    the samples and the second in about 4 %."""
    tint = np.float32([0.5, 1.0, 0.25])
    osc = _slab_scene(1.5, tint)
    osc.material_model = (0, 0, 1.0, 1)
    cam = oracle.default_camera(position=(-0.5, -0.5, 5.0), forward=(0, 0, -1), vfov_rad=0.0)
    st = oracle.default_settings(ray_bounce_limit=6, tone_mapping=0, gamma_correction=0)
    sky = lambda dy: ((np.float32(1.0) - np.float32(0.5 * (1 + dy))) * np.float32(1) + np.float32(0.5 * (1 + dy)) * np.float32(st.sky_color)) ** 2 * np.float32(st.sky_intensity)
    through, mirrored = sky(0.0) * tint * tint, sky(0.0) * tint
    n_through = n_mirrored = n_other = 0
    for frame in range(1, 41):
        _, acc, _ = oracle.render(osc, cam, st, 8, 8, frame, 1)           # (accumulation buffer of ONE frame = the samples themselves)
        for px in acc.reshape(-1, 3):
            if np.allclose(px, through, rtol=1e-5): n_through += 1
            elif np.allclose(px, mirrored, rtol=1e-5): n_mirrored += 1
            else: n_other += 1
    total = n_through + n_mirrored + n_other
    assert 0.88 < n_through / total < 0.96, (n_through, n_mirrored, n_other)
    assert 0.01 < n_mirrored / total < 0.08, (n_through, n_mirrored, n_other)
    # switched off, the same material is the reference's diffuse surface: the image differs
    osc.material_model = (0, 0, 1.0, 0)
    _, off, _ = oracle.render(osc, cam, st, 8, 8, 1, 1)
    osc.material_model = (0, 0, 1.0, 1)
    _, on, _ = oracle.render(osc, cam, st, 8, 8, 1, 1)
    assert not np.array_equal(bits(on), bits(off))


def test_oracle_dielectric_total_internal_reflection():
    """Hand-derived.  A camera INSIDE the slab (z = 0.5) looking up at 60 degrees from the normal: leaving glass of index 1.5 the
    critical angle is asin(1 / 1.5) = 41.8 degrees, so ri * sin = 1.5 * 0.866 > 1 and the ray is reflected at the top face, then at
    the bottom face, and so on: it never leaves, whatever the random numbers (Random.cu:26-40 is not even asked).  With the bounce
    limit reached inside the slab the sample is black.  At 30 degrees (1.5 * 0.5 < 1) the ray gets out: light."""
    osc = _slab_scene(1.5)
    osc.material_model = (0, 0, 1.0, 1)
    st = oracle.default_settings(ray_bounce_limit=12, tone_mapping=0, gamma_correction=0)
    s60, c60 = np.sin(np.radians(60.0)), np.cos(np.radians(60.0))
    trapped = oracle.default_camera(position=(-30.0, -5.0, 0.5), forward=(s60, 0, c60), vfov_rad=0.0)
    for frame in (1, 2, 3, 7):
        _, acc, _ = oracle.render(osc, trapped, st, 4, 4, frame, 1)
        assert not acc.any(), "a ray beyond the critical angle left the slab"
    s30, c30 = np.sin(np.radians(30.0)), np.cos(np.radians(30.0))
    free = oracle.default_camera(position=(-30.0, -5.0, 0.5), forward=(s30, 0, c30), vfov_rad=0.0)
    lit = 0
    for frame in range(1, 9):
        _, acc, _ = oracle.render(osc, free, st, 4, 4, frame, 1)
        lit += int((acc.reshape(-1, 3).max(axis=1) > 0).sum())
    assert lit > 100                                                       # 128 samples, ~4-10 % reflected back in


def _renderer(kernel):
    import os
    old = os.environ.get("DRT_KERNEL")
    os.environ["DRT_KERNEL"] = kernel
    try:
        return drt.Renderer(0)
    finally:
        if old is None: os.environ.pop("DRT_KERNEL", None)
        else: os.environ["DRT_KERNEL"] = old


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["path_pool", "wave_queue"])
@pytest.mark.parametrize("name", ["emissive_test", "cornell_box_gltf", "cs16_dust"])
@pytest.mark.parametrize("model", [(1, 0, 1.0), (0, 1, 1.0), (1, 1, 2.5)])
def test_material_model_matches_the_oracle_and_is_off_by_default(name, model, kernel):
    """On: the production kernel (path_pool, material-model variants: scene in LDS / read from global memory, with and without
    sunlight) and the general wave_queue kernel == the oracle with the same switches, bit for bit (EmissiveTest.glb: emissive cubes
    and a mirror sphere; cornell_box.gltf: the ceiling light mesh; cs16_dust: every glTF-default material is 'metallic', roughness 1).
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
        r = _renderer(kernel)
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
        assert ("path_pool<materials" if kernel == "path_pool" else "general") in r.kernelInfo(), r.kernelInfo()
        osc.material_model = model
        ref, _, _ = oracle.render(osc, ocam, st, W, H, 1, frames)
        osc.material_model = (0, 0, 1.0)
        nbad = int((bits(img) != bits(ref)).any(axis=-1).sum())
        assert nbad == 0, "%s %r sun=%d %s: %d pixels differ from the oracle" % (name, model, sun, r.kernelInfo(), nbad)
        mats = sc.m_Material
        touches = (model[0] and np.asarray(mats["emissive"]).any()) or (model[1] and np.asarray(mats["metallic"]).any())
        if touches:                                                          # (a scene without such materials looks the same)
            assert not np.array_equal(bits(img), bits(plain))                # the switch does change the image
        else:
            assert np.array_equal(bits(img), bits(plain))
        r.setMaterialModel(0, 0, 1.0)
        r.resetAccumulationBuffer()
        r.RenderBatch(cam, sc, frames)
        assert "materials" not in r.kernelInfo()
        assert np.array_equal(bits(r.GetRenderTargetImage()), bits(plain))


def _glass_scene(n_extra=0):
    """A room of six quads (diffuse, one emissive, one mirror) with a glass box and a glass 'lens' pane inside; the same arrays go to
    the product (drt_scene_set_geometry / add_material_ex) and to the oracle.  n_extra > 0 adds that many small diffuse triangles on
    the floor, so that the traversal data outgrows LDS and the hbm-scene build of the kernel renders it."""
    def quad(a, b, c, d, n):
        return [a, b, c, a, c, d], [n] * 6
    P, N, M = [], [], []
    def add(a, b, c, d, n, m):
        p, nn = quad(a, b, c, d, n); P.extend(p); N.extend(nn); M.extend([m, m])
    R = 3.0
    add((-R, 0, -R), (R, 0, -R), (R, 0, R), (-R, 0, R), (0, 1, 0), 0)                 # floor
    add((-R, 4, -R), (-R, 4, R), (R, 4, R), (R, 4, -R), (0, -1, 0), 1)                # ceiling (emissive)
    add((-R, 0, -R), (-R, 4, -R), (R, 4, -R), (R, 0, -R), (0, 0, 1), 0)               # back wall
    add((-R, 0, -R), (-R, 0, R), (-R, 4, R), (-R, 4, -R), (1, 0, 0), 2)               # left wall (mirror)
    add((R, 0, -R), (R, 4, -R), (R, 4, R), (R, 0, R), (-1, 0, 0), 0)                  # right wall
    lo, hi = np.float32([-0.8, 0.4, -1.2]), np.float32([0.6, 1.8, 0.2])               # glass box, outward normals
    x0, y0, z0 = lo; x1, y1, z1 = hi
    add((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1), (0, 0, 1), 3)
    add((x1, y0, z0), (x0, y0, z0), (x0, y1, z0), (x1, y1, z0), (0, 0, -1), 3)
    add((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0), (-1, 0, 0), 3)
    add((x1, y0, z1), (x1, y0, z0), (x1, y1, z0), (x1, y1, z1), (1, 0, 0), 3)
    add((x0, y1, z1), (x1, y1, z1), (x1, y1, z0), (x0, y1, z0), (0, 1, 0), 3)
    add((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1), (0, -1, 0), 3)
    add((1.2, 0.3, 0.9), (2.4, 0.3, 0.3), (2.4, 2.2, 0.3), (1.2, 2.2, 0.9), (0.4472136, 0, 0.8944272), 4)    # a thin tinted pane, seen from both sides
    rng = np.random.default_rng(7)
    for _ in range(n_extra):
        c = np.float32([rng.uniform(-2.8, 2.8), 0.002, rng.uniform(-2.8, 2.8)])
        d = np.float32(rng.uniform(-0.05, 0.05, (3, 3))); d[:, 1] = 0
        P.extend([tuple(c + d[0]), tuple(c + d[1]), tuple(c + d[2])]); N.extend([(0, 1, 0)] * 3); M.append(0)
    pos = np.float32(P).reshape(-1, 3, 3); nrm = np.float32(N).reshape(-1, 3, 3); uv = np.zeros((len(M), 3, 2), np.float32)
    mat = np.int32(M)
    # (albedo, emissive, roughness, metallic, transmission, refractive index)
    mats = [((0.7, 0.7, 0.7), (0, 0, 0), 0.5, 0, 0, 1.45), ((0.9, 0.9, 0.9), (6.0, 5.0, 4.0), 0.5, 0, 0, 1.45), ((0.9, 0.9, 0.95), (0, 0, 0), 0.05, 1, 0, 1.45),
            ((1.0, 1.0, 1.0), (0, 0, 0), 0.0, 0, 1, 1.5), ((0.6, 0.9, 0.7), (0, 0, 0), 0.0, 1, 1, 1.33)]
    sc = drt.Scene()
    for alb, em, rough, metal, trans, ior in mats:
        sc.addMaterialEx(alb, -1, em, rough, bool(metal), bool(trans), ior)
    sc.setGeometry(pos, nrm, uv, mat)
    b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 4, 8; b.buildIterative(sc)
    tris = np.zeros(len(mat), oracle.TRI_DTYPE)
    a = [np.ascontiguousarray(x, np.float32) for x in (pos.reshape(-1, 3), nrm.reshape(-1, 3), uv.reshape(-1, 2))]
    oracle.lib().o_build_triangles(a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, mat.ctypes.data, len(mat), tris.ctypes.data)
    osc = oracle.Scene(tris, [(np.float32(m[0]), -1) for m in mats], [],
                       [(np.float32(m[1]), np.float32(m[2]), m[3], m[4], np.float32(m[5])) for m in mats]).build_bvh(4, 8)
    return sc, osc


@pytest.mark.gpu
@pytest.mark.parametrize("n_extra", [0, 3000])
@pytest.mark.parametrize("model", [(0, 0, 1.0, 1), (1, 1, 1.5, 1), (1, 0, 1.0, 1)])
def test_dielectric_lobe_on_the_production_kernel(model, n_extra):
    """The dielectric lobe (drt_material_model.transmission; refract / reflectance of CudaMath/Random.cu:26-40) on path_pool -- scene in
    LDS (n_extra = 0) and read from global memory (3 000 more triangles), with and without sunlight, alone and together with the emissive
    term and the mirror lobe -- against the oracle's statement of the same rule, bit for bit; a material that is both metallic and
    transmissive is glass (precedence); off, the glass is the reference's diffuse surface.  PARITY UNPINNED: the rule is this
    library's own (the reference loads the fields and never reads them); only the two hand-derived oracle cases above are outside checks."""
    sc, osc = _glass_scene(n_extra)
    pos, fwd = (0.3, 1.6, 2.8), (-0.1, -0.15, -1.0)
    cam = drt.Camera(pos); cam.m_Forward_dir = np.array(fwd, np.float32)
    ocam = oracle.default_camera(position=pos, forward=fwd)
    W, H, frames, depth = 160, 90, 3, 7
    r = drt.Renderer(0)
    for sun in (0, 1):
        st = oracle.default_settings(ray_bounce_limit=depth, enable_sunlight=sun)
        r.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=depth, enableSunlight=sun)
        r.ResizeBuffer(W, H)
        r.setMaterialModel(0, 0, 1.0, 0)
        r.resetAccumulationBuffer()
        r.RenderBatch(cam, sc, frames)
        plain = r.GetRenderTargetImage()
        assert ("hbm-scene" if n_extra else "lds-scene") in r.kernelInfo() and "materials" not in r.kernelInfo(), r.kernelInfo()
        ref_plain, _, _ = oracle.render(osc, ocam, st, W, H, 1, frames)
        assert np.array_equal(bits(plain), bits(ref_plain))
        r.setMaterialModel(*model)
        r.resetAccumulationBuffer()
        r.RenderBatch(cam, sc, frames)
        img = r.GetRenderTargetImage()
        assert "path_pool<materials" in r.kernelInfo(), r.kernelInfo()
        osc.material_model = model
        ref, _, _ = oracle.render(osc, ocam, st, W, H, 1, frames)
        osc.material_model = (0, 0, 1.0)
        nbad = int((bits(img) != bits(ref)).any(axis=-1).sum())
        assert nbad == 0, "model %r sun=%d %s: %d pixels differ from the oracle" % (model, sun, r.kernelInfo(), nbad)
        assert not np.array_equal(bits(img), bits(plain))
    # the general wave_queue kernel does not know the dielectric lobe: refused, not silently ignored
    rq = _renderer("wave_queue")
    rq.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=depth)
    rq.ResizeBuffer(W, H)
    rq.setMaterialModel(0, 0, 1.0, 1)
    with pytest.raises(drt.DrtError):
        rq.RenderBatch(cam, sc, 1)
