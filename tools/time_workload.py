"""GPU box: time RenderBatch for a scene at a given size (python tools/time_workload.py scene W H spp [depth])."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dustraytracer_amd as drt
from tests.scenes import SCENES, scene_path
name, W, H, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
_, pos, fwd, depth = SCENES[name]
if len(sys.argv) > 5: depth = int(sys.argv[5])
sc = drt.Scene(); sc.loadGLTFmodel(scene_path(name))
b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8; b.buildIterative(sc)
cam = drt.Camera(pos); cam.m_Forward_dir = np.array(fwd, np.float32)
r = drt.Renderer(0)
extra = dict(enableSunlight=1) if os.environ.get("DRT_TW_SUN") == "1" else {}
r.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=depth, max_samples=spp + 1, **extra)
r.ResizeBuffer(W, H)
for k in range(2):
    r.resetAccumulationBuffer(); t0 = time.time(); ms = r.RenderBatch(cam, sc, spp); wall = time.time() - t0
    print(name, W, H, spp, "depth", depth, r.kernelInfo(), "ms %.3f wall %.3f s  %.1f Msamples/s" % (ms, wall, W * H * spp / ms / 1e3), flush=True)
