"""GPU box: phase statistics of the wave_queue kernel on a workload."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dustraytracer_amd as drt
from tests.scenes import SCENES, scene_path
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
W, H, spp = 1920, 1080, int(sys.argv[2]) if len(sys.argv) > 2 else 8
_, pos, fwd, depth = SCENES[name]
sc = drt.Scene(); sc.loadGLTFmodel(scene_path(name))
b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8; b.buildIterative(sc)
cam = drt.Camera(pos); cam.m_Forward_dir = np.array(fwd, np.float32)
r = drt.Renderer(0)
r.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=depth, max_samples=spp + 1)
r.ResizeBuffer(W, H)
r.setCounting(True); ms = r.RenderBatch(cam, sc, spp); c = r.getCounters(); r.setCounting(False)
print(r.kernelInfo(), "counting ms %.2f" % ms, c.as_dict())
ps = c.phase_stats(); print("phase stats", ps)
tot = sum(e for e, _ in ps.values())
print("execs (M): " + "  ".join("%s %.2f @ %.1f lanes" % (k, ps[k][0] / 1e6, ps[k][1]) for k in "TNSR"))
tt = [int(c.phase_ticks[i]) for i in range(4)]; wt = int(c.wave_ticks)
print("wave time shares: " + "  ".join("%s %.1f%%" % (k, 100.0 * tt[i] / wt) for i, k in enumerate("TNSR")) + "  other %.1f%%" % (100.0 * (wt - sum(tt)) / wt))
print("ticks per phase run: " + "  ".join("%s %.0f" % (k, tt[i] / max(int(c.phase_execs[i]), 1)) for i, k in enumerate("TNSR")))
for k in range(3):
    r.resetAccumulationBuffer(); ms = r.RenderBatch(cam, sc, spp)
print(r.kernelInfo(), "ms %.3f  %.1f Msamples/s" % (ms, W * H * spp / ms / 1e3))
