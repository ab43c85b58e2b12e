import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np, dustraytracer_amd as drt
from tests.scenes import SCENES, scene_path
name = "cornell_box"
_, pos, fwd, _ = SCENES[name]
sc = drt.Scene(); sc.loadGLTFmodel(scene_path(name))
b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8; b.buildIterative(sc)
cam = drt.Camera(pos); cam.m_Forward_dir = np.array(fwd, np.float32)
r = drt.Renderer(0)
for H in (8, 16, 32, 64, 136, 272, 544, 1080):
    for spp in (8,):
        r.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=8, max_samples=spp + 1)
        r.ResizeBuffer(1920, H)
        spans, evs = [], []
        for k in range(6):
            r.resetAccumulationBuffer(); ms = r.RenderBatch(cam, sc, spp)
            if k >= 2: spans.append(r.kernelSpanMs()); evs.append(ms)
        n = 1920 * H * spp / 1e6
        print("1920x%-4d x%d: %.2f Msamples  kernel span %.4f ms  events %.4f ms   %.0f Msamples/s (span)  %s" % (H, spp, n, np.mean(spans), np.mean(evs), n / np.mean(spans) * 1e3, r.kernelInfo()), flush=True)
