"""GPU box: kernel time (HIP events, isolated launches) against the number of samples, to split the per-launch fixed cost
from the per-sample cost.  python tools/fixed_cost.py [scene]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dustraytracer_amd as drt
from tests.scenes import SCENES, scene_path
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
_, pos, fwd, depth = SCENES[name]
depth = 8
sc = drt.Scene(); sc.loadGLTFmodel(scene_path(name))
b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8; b.buildIterative(sc)
cam = drt.Camera(pos); cam.m_Forward_dir = np.array(fwd, np.float32)
r = drt.Renderer(0)
pts = []
for (W, H, spp) in [(64, 64, 1), (256, 256, 1), (512, 512, 1), (1920, 1080, 1), (1920, 136, 8), (1920, 1080, 2), (1920, 1080, 4), (1920, 1080, 8)]:
    r.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=depth, max_samples=spp + 1)
    r.ResizeBuffer(W, H)
    ms = []
    for k in range(40):
        r.resetAccumulationBuffer(); ms.append(r.RenderBatch(cam, sc, spp))
    ms = sorted(ms[5:]); med = ms[len(ms) // 2]
    n = W * H * spp
    pts.append((n, med))
    print("%5dx%-5d spp %d  samples %9d  kernel_ms median %.4f  min %.4f  -> %.1f Ms/s" % (W, H, spp, n, med, ms[0], n / med / 1e3), flush=True)
x = np.array([p[0] for p in pts], float); y = np.array([p[1] for p in pts], float)
A = np.vstack([np.ones_like(x), x]).T
a, bb = np.linalg.lstsq(A, y, rcond=None)[0]
print("fit: ms = %.4f + %.4f per Msample  (asymptote %.1f Ms/s)" % (a, bb * 1e6, 1e3 / (bb * 1e6) if bb > 0 else 0))
