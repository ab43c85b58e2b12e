"""GPU box: randomised parity.  Random scene, camera (inside, outside, far away, degenerate directions), lens, settings,
frame size and frame count; every case compared bit for bit with the CPU oracle.
   python tools/fuzz_parity.py [cases] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dustraytracer_amd as drt
import oracle
from tests.scenes import SCENES, scene_path

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
names = sorted(SCENES) + (["glass_lds", "glass_hbm"] * 2 if os.environ.get("DRT_FUZZ_MATERIALS") == "1" else [])
loaded = {}


def get(name):
    if name not in loaded and name.startswith("glass"):
        # the programmatic room with glass, mirror and emissive materials of tests/test_material_model.py (scene in LDS / read from global memory)
        from tests.test_material_model import _glass_scene
        sc, osc = _glass_scene(0 if name == "glass_lds" else 3000)
        p = osc.tris["p"].reshape(-1, 3)
        loaded[name] = (sc, osc, p.min(0), p.max(0))
    if name not in loaded:
        sc = drt.Scene(); sc.loadGLTFmodel(scene_path(name))
        b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8; b.buildIterative(sc)
        osc = oracle.Scene.load_glb(scene_path(name)).build_bvh(20, 8)
        p = osc.tris["p"].reshape(-1, 3)
        loaded[name] = (sc, osc, p.min(0), p.max(0))
    return loaded[name]


def renderer_with(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return drt.Renderer(0)                  # the knobs are read when the renderer is created
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


# the launcher's own choice of how many stack levels stay in LDS, and 1 / 3 levels forced (deep trees: the rest of a stack lives in HBM)
renderers = [drt.Renderer(0), renderer_with({"DRT_POOL_STACK_LDS": "1"}), renderer_with({"DRT_POOL_STACK_LDS": "3"})]
bad, kernels, t0 = 0, {}, time.time()
for case in range(n_cases):
    r = renderers[0 if rng.integers(2) == 0 else int(rng.integers(1, 3))]
    name = names[rng.integers(len(names))]
    sc, osc, lo, hi = get(name)
    ext = hi - lo
    kind = rng.integers(6)
    if kind == 0:   pos = lo + rng.uniform(0, 1, 3) * ext                                   # inside the bounds
    elif kind == 1: pos = (lo + hi) / 2 + rng.normal(size=3) * ext                          # around
    elif kind == 2: pos = (lo + hi) / 2 + rng.normal(size=3) * ext * 50                     # far away
    elif kind == 3: pos = lo + np.round(rng.uniform(0, 1, 3) * 4) / 4 * ext                 # on a lattice: often exactly on geometry / box planes
    elif kind == 4: pos = np.array(SCENES[name][1] if name in SCENES else (0.3, 1.6, 2.8), np.float64)                          # the benchmark pose
    else:           pos = hi + ext * 0.01
    fwd = rng.normal(size=3)
    if rng.integers(4) == 0: fwd = np.eye(3)[rng.integers(3)] * rng.choice([-1, 1])         # axis-aligned: zero components -> inf inverse
    if rng.integers(5) == 0: fwd = (lo + hi) / 2 - pos                                      # look at the scene
    if not np.any(fwd): fwd = np.array([0, 0, -1.0])
    pos = pos.astype(np.float32); fwd = fwd.astype(np.float32)
    cam_kw = {}
    if rng.integers(3) == 0: cam_kw.update(defocus_angle=float(rng.uniform(0.1, 3)), focus_dist=float(rng.uniform(0.2, 30)))
    if rng.integers(3) == 0: cam_kw.update(vfov_rad=float(rng.uniform(0.2, 2.6)))
    if rng.integers(4) == 0: cam_kw.update(exposure=float(rng.uniform(0.2, 4)))
    st = dict(ray_bounce_limit=int(rng.integers(0, 13)))
    if rng.integers(3) == 0: st["enableSunlight"] = 1
    if rng.integers(5) == 0: st["tone_mapping"] = 0
    if rng.integers(5) == 0: st["gamma_correction"] = 0
    if rng.integers(6) == 0: st.update(RenderMode=1, DebugMode=int(rng.integers(0, 5)))
    if rng.integers(5) == 0: st.update(sky_intensity=float(rng.uniform(0, 40)), sunlight_intensity=float(rng.uniform(0, 60)),
                                       sunlight_dir=(float(rng.uniform(-1, 1)), float(rng.uniform(0, 1))))
    W, H, frames = int(rng.integers(1, 97)), int(rng.integers(1, 65)), int(rng.integers(1, 4))
    # DRT_FUZZ_MATERIALS=1: the opt-in material model too (emissive term, mirror lobe, dielectric lobe), any combination, in a third of the cases
    model = (0, 0, 1.0, 0)
    if os.environ.get("DRT_FUZZ_MATERIALS") == "1" and (name.startswith("glass") or rng.integers(3) == 0):
        model = (int(rng.integers(2)), int(rng.integers(2)), float(rng.uniform(0.1, 3.0)), int(rng.integers(2)))
    if "RenderMode" in st: model = (model[0], model[1], model[2], 0)                          # (debug views go to wave_queue, which has no dielectric lobe)
    if st.get("ray_bounce_limit", 0) > 12: pass
    cam = drt.Camera(pos); cam.m_Forward_dir = fwd
    ocam = oracle.default_camera(position=tuple(float(v) for v in pos), forward=tuple(float(v) for v in fwd))
    for k, v in cam_kw.items():
        setattr(cam, k, v); setattr(ocam, k, v)
    oname = {"enableSunlight": "enable_sunlight", "RenderMode": "render_mode", "DebugMode": "debug_mode"}
    r.m_RendererSettings = drt.RendererSettings(max_samples=frames + 1, **st)
    r.ResizeBuffer(W, H); r.resetAccumulationBuffer()
    r.setMaterialModel(*model)
    osc.material_model = model
    r.RenderBatch(cam, sc, frames)
    img = r.GetRenderTargetImage()
    ref, _, _ = oracle.render(osc, ocam, oracle.default_settings(**{oname.get(k, k): v for k, v in st.items()}), W, H, 1, frames, threads=8)
    osc.material_model = (0, 0, 1.0, 0)
    kname = r.kernelInfo().split()[0] + (" deep stacks" if " in LDS)" in r.kernelInfo() else "")
    kernels[kname] = kernels.get(kname, 0) + 1
    a, b = img.view(np.uint32), ref.view(np.uint32)
    both_nan = np.isnan(img) & np.isnan(ref)
    nbad = int(((a != b) & ~both_nan).any(axis=-1).sum())
    if nbad:
        bad += 1
        print("MISMATCH case %d: %s pos %s fwd %s cam %s settings %s model %s %dx%d x%d: %d pixels" % (case, name, pos.tolist(), fwd.tolist(), cam_kw, st, model, W, H, frames, nbad), flush=True)
    if case % 50 == 49:
        print("... %d cases, %d mismatching, %.0f s" % (case + 1, bad, time.time() - t0), flush=True)
print("cases %d, mismatching %d; kernels used: %s" % (n_cases, bad, kernels))
sys.exit(1 if bad else 0)
