"""GPU box: Render() (one frame per call) at 1080p on cornell under several path_pool settings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
import dustraytracer_amd as drt
from tests.scenes import SCENES, scene_path
from tools.pool_check import renderer
name, W, H, depth = "cornell_box", 1920, 1080, 8
sc = drt.Scene(); sc.loadGLTFmodel(scene_path(name))
b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8; b.buildIterative(sc)
_, pos, fwd, _ = SCENES[name]
cam = drt.Camera(pos); cam.m_Forward_dir = np.array(fwd, np.float32)
for spec in sys.argv[1:]:
    env = {"DRT_KERNEL": "path_pool"}
    for kv in spec.split(","):
        if "=" in kv:
            k, v = kv.split("="); env["DRT_POOL_" + k] = v
    r = renderer(env)
    r.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=depth, max_samples=100000)
    r.ResizeBuffer(W, H)
    for _ in range(10): r.Render(cam, sc)
    t0 = time.perf_counter(); span = 0.0
    for _ in range(200):
        r.Render(cam, sc); span += r.kernelSpanMs()
    wall = (time.perf_counter() - t0) / 200 * 1e3
    print("%-40s wall %.3f ms/call span %.3f [%s]" % (spec, wall, span / 200, r.kernelInfo()), flush=True)
