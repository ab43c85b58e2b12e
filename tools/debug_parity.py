"""Debug helper (GPU box): per-debug-mode mismatch counts between the HIP path and the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dustraytracer_amd as drt, oracle
from tests.scenes import SCENES, scene_path, bits

name = sys.argv[1] if len(sys.argv) > 1 else "dense_monkey"
W, H = 160, 90
_, pos, fwd, depth = SCENES[name]
sc = drt.Scene(); sc.loadGLTFmodel(scene_path(name))
b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8; b.buildIterative(sc)
osc = oracle.Scene.load_glb(scene_path(name)).build_bvh(20, 8)
cam = drt.Camera(pos); cam.m_Forward_dir = np.array(fwd, np.float32)
ocam = oracle.default_camera(position=pos, forward=fwd)
r = drt.Renderer(0)
for rm, dm, bl in [(1,0,2),(1,1,2),(1,2,2),(1,3,2),(1,4,2),(0,0,0),(0,0,1),(0,0,2)]:
    r.m_RendererSettings = drt.RendererSettings(RenderMode=rm, DebugMode=dm, ray_bounce_limit=bl, tone_mapping=0, gamma_correction=0)
    r.ResizeBuffer(W, H); r.resetAccumulationBuffer(); r.RenderBatch(cam, sc, 1)
    img = r.GetRenderTargetImage()
    ref, _, _ = oracle.render(osc, ocam, oracle.default_settings(render_mode=rm, debug_mode=dm, ray_bounce_limit=bl, tone_mapping=0, gamma_correction=0), W, H, 1, 1)
    bad = (bits(img) != bits(ref)).any(-1)
    print("render_mode", rm, "debug_mode", dm, "bounces", bl, "mismatch px", int(bad.sum()), "Linf %.3e" % np.abs(img-ref).max())
    if bad.any():
        ys, xs = np.nonzero(bad)
        for y, x in list(zip(ys, xs))[:4]:
            print("   px", x, y, "gpu", img[y, x, :3], "ref", ref[y, x, :3])
