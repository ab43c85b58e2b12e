"""GPU box: what the opt-in material model costs, and what moving it from the general wave_queue kernel (round 2) to path_pool (round 3) bought.
   python tools/material_model_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dustraytracer_amd as drt
from tests.scenes import SCENES, scene_path


def renderer(kernel):
    old = os.environ.get("DRT_KERNEL")
    os.environ["DRT_KERNEL"] = kernel
    try:
        return drt.Renderer(0)
    finally:
        os.environ.pop("DRT_KERNEL", None) if old is None else os.environ.__setitem__("DRT_KERNEL", old)


W, H, spp = 1920, 1080, 8
for name in ("cornell_box_gltf", "emissive_test", "cs16_dust"):
    sc = drt.Scene(); sc.loadGLTFmodel(scene_path(name))
    b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8; b.buildIterative(sc)
    _, pos, fwd, depth = SCENES[name]
    cam = drt.Camera(pos); cam.m_Forward_dir = np.array(fwd, np.float32)
    for kernel in ("path_pool", "wave_queue"):
        r = renderer(kernel)
        for model in ((0, 0, 1.0), (1, 1, 1.0)):
            r.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=depth, max_samples=spp + 1)
            r.ResizeBuffer(W, H)
            r.setMaterialModel(*model)
            best = 1e9
            for k in range(4):
                r.resetAccumulationBuffer(); ms = r.RenderBatch(cam, sc, spp)
                if k: best = min(best, ms)
            print("%-18s %dx%d x %d spp depth %d  model %s  %-60s %.3f ms  %.0f Msamples/s" % (name, W, H, spp, depth, "on " if model[0] else "off", r.kernelInfo(), best, W * H * spp / best / 1e3), flush=True)
