"""GPU box: the interactive call (one frame index per Render, Renderer.cu:80-117) and other small launches, both kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dustraytracer_amd as drt
from tests.scenes import SCENES, scene_path
from tools.pool_check import renderer

def one(name, W, H, depth, kernel, calls=200):
    sc = drt.Scene(); sc.loadGLTFmodel(scene_path(name))
    b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8; b.buildIterative(sc)
    _, pos, fwd, _ = SCENES[name]
    cam = drt.Camera(pos); cam.m_Forward_dir = np.array(fwd, np.float32)
    r = renderer({"DRT_KERNEL": kernel})
    r.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=depth, max_samples=100000)
    r.ResizeBuffer(W, H)
    for _ in range(10): r.Render(cam, sc)
    t0 = time.perf_counter(); ms = 0.0; span = 0.0
    for _ in range(calls):
        ms += r.Render(cam, sc); span += r.kernelSpanMs()
    wall = (time.perf_counter() - t0) / calls * 1e3
    print("%-14s %4dx%-4d d%-2d Render() x%d  %-12s wall %.3f ms/call  events %.3f  kernel span %.3f  -> %.1f Msamples/s  [%s]" % (
        name, W, H, depth, calls, kernel, wall, ms / calls, span / calls, W * H / wall / 1e3, r.kernelInfo()), flush=True)

for k in ("path_pool", "wave_queue"):
    one("cornell_box", 1920, 1080, 8, k)
    one("cornell_box", 256, 256, 4, k)
    one("room", 1920, 1080, 16, k, calls=50)
    one("suzanne_plane", 1920, 1080, 2, k)
