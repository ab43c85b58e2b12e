"""GPU box: what the gather of drt_group_render_batch costs per frame, both ways, with the one device sending to itself
(DRT_GROUP_FORCE_RCCL=1): one ncclSend / ncclRecv pair for the whole shard + the assemble pass (default), against one pair per
8-row stripe received in place (DRT_GROUP_GATHER=stripes, round 2).  A group of one without RCCL is the baseline (render only +
the assemble pass).  usage: python tools/group_gather_bench.py [W H spp steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
W, H, spp, steps = (int(v) for v in (sys.argv[1:5] + ["1920", "1080", "1", "100"][len(sys.argv) - 1:]))
import dustraytracer_amd as drt
from tests.scenes import SCENES, scene_path
sc = drt.Scene(); sc.loadGLTFmodel(scene_path("cornell_box"))
b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8; b.buildIterative(sc)
_, pos, fwd, _ = SCENES["cornell_box"]
cam = drt.Camera(pos); cam.m_Forward_dir = np.array(fwd, np.float32)
for label, env in (("no RCCL (render + assemble)", {}), ("whole shard, one pair", {"DRT_GROUP_FORCE_RCCL": "1"}),
                   ("per stripe, %d pairs" % ((H + 7) // 8), {"DRT_GROUP_FORCE_RCCL": "1", "DRT_GROUP_GATHER": "stripes"})):
    old = {k: os.environ.get(k) for k in ("DRT_GROUP_FORCE_RCCL", "DRT_GROUP_GATHER")}
    os.environ.update(env)
    try:
        g = drt.RendererGroup([0])
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    g.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=8, max_samples=spp + 1)
    g.ResizeBuffer(W, H)
    for i in range(10):
        g.resetAccumulationBuffer(); g.RenderBatch(cam, sc, spp)
    t0 = time.perf_counter()
    for i in range(steps):
        g.resetAccumulationBuffer(); g.RenderBatch(cam, sc, spp)
    dt = (time.perf_counter() - t0) / steps
    print("%-32s %dx%d x %d spp: %.3f ms per frame (blocking steps)" % (label, W, H, spp, dt * 1e3), flush=True)
    del g
