"""GPU box: path_pool under the bench's conditions (several renderers in flight on their own streams), statistics build with its
range checks on every global index (DRT_POOL_STATS=1): any check that fires, or any frame that differs from the first, is reported.
   python tools/pool_stress.py scene W H spp depth steps [frames_in_flight]"""
import os, sys
os.environ.setdefault("DRT_POOL_STATS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dustraytracer_amd as drt
from tests.scenes import SCENES, scene_path

name, W, H, spp, depth, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
fif = int(sys.argv[7]) if len(sys.argv) > 7 else 3
_, pos, fwd, _ = SCENES[name]
sc = drt.Scene(); sc.loadGLTFmodel(scene_path(name))
b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8; b.buildIterative(sc)
cam = drt.Camera(pos); cam.m_Forward_dir = np.array(fwd, np.float32)
dev = torch.device("cuda", 0)
slots = []
for k in range(fif):
    r = drt.Renderer(0)
    r.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=depth, max_samples=spp + 1)
    r.setFramesInFlight(fif)
    r.ResizeBuffer(W, H)
    st = torch.cuda.Stream(device=dev)
    r.setStream(st.cuda_stream)
    slots.append([r, st, False])
first = None
bad = 0
for i in range(steps):
    r, st, busy = slots[i % fif]
    if busy:
        r.Wait()
        img = r.GetRenderTargetImage()
        if first is None: first = img.copy()
        elif not np.array_equal(img.view(np.uint32), first.view(np.uint32)): bad += 1
    with torch.cuda.stream(st):
        r.resetAccumulationBuffer()
        r.RenderBatchAsync(cam, sc, spp)
    slots[i % fif][2] = True
for r, st, busy in slots:
    if busy: r.Wait()
print(name, W, H, spp, "depth", depth, slots[0][0].kernelInfo(), "steps", steps, "in flight", fif, "frames differing from the first:", bad)
