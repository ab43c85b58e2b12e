"""GPU box: large whole-frame parity runs that are too long for the test suite.  Every pixel of every frame is compared
bit for bit with the CPU oracle.   python tools/soak_parity.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dustraytracer_amd as drt
import oracle
from tests.scenes import SCENES, scene_path

CASES = [("cornell_box", 1920, 1080, 64, 8, {}), ("cs16_dust", 1920, 1080, 8, 5, {}), ("mc_transparency", 1920, 1080, 16, 5, {}),
         ("suzanne_plane", 3840, 2160, 8, 4, {}), ("sunshadow_test", 1920, 1080, 8, 3, dict(enableSunlight=1)),
         ("room", 1920, 1080, 8, 16, {}), ("lightweight_rt", 1920, 1080, 16, 6, dict(enableSunlight=1)),
         ("dense_monkey", 1920, 1080, 8, 8, dict(tone_mapping=0)),
         # round 2: sunlight and cut-outs on path_pool
         ("cornell_box", 1920, 1080, 8, 8, dict(enableSunlight=1)), ("room", 1920, 1080, 4, 16, dict(enableSunlight=1)),
         ("uv_texture_test", 1920, 1080, 8, 5, dict(enableSunlight=1)), ("uv_texture_test", 1920, 1080, 8, 5, {})]
if len(sys.argv) > 1 and sys.argv[1] == "c5":           # BASELINE config 5 whole: 531 M samples, several minutes of oracle time
    CASES = [("room", 3840, 2160, 64, 16, {})]
names = {"enableSunlight": "enable_sunlight"}
r = drt.Renderer(0)
bad_total = 0
for name, W, H, spp, depth, kw in CASES:
    _, pos, fwd, _ = SCENES[name]
    sc = drt.Scene(); sc.loadGLTFmodel(scene_path(name))
    b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8; b.buildIterative(sc)
    cam = drt.Camera(pos); cam.m_Forward_dir = np.array(fwd, np.float32)
    r.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=depth, max_samples=spp + 1, **kw)
    r.ResizeBuffer(W, H); r.resetAccumulationBuffer()
    ms = r.RenderBatch(cam, sc, spp)
    img = r.GetRenderTargetImage()
    osc = oracle.Scene.load_glb(scene_path(name)).build_bvh(20, 8)
    t0 = time.time()
    ref, _, _ = oracle.render(osc, oracle.default_camera(position=pos, forward=fwd),
                              oracle.default_settings(ray_bounce_limit=depth, **{names.get(k, k): v for k, v in kw.items()}), W, H, 1, spp)
    dt = time.time() - t0
    nbad = int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
    bad_total += nbad
    print("%-16s %dx%d x%d spp depth %d %s: %s  GPU %.1f ms (%.0f Msamples/s), oracle %.1f s, pixels not bit-equal: %d"
          % (name, W, H, spp, depth, kw, r.kernelInfo(), ms, W * H * spp / ms / 1e3, dt, nbad), flush=True)
print("TOTAL pixels not bit-equal:", bad_total)
sys.exit(1 if bad_total else 0)
