"""GPU box: path_pool against the oracle (bit-exact?) and against wave_queue (time), a few scenes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import dustraytracer_amd as drt
from tests.scenes import SCENES, scene_path, bits


def renderer(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return drt.Renderer(0)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def run(name, W, H, frames, depth, env, check=True, reps=3):
    sc = drt.Scene(); sc.loadGLTFmodel(scene_path(name))
    b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8; b.buildIterative(sc)
    _, pos, fwd, _ = SCENES[name]
    cam = drt.Camera(pos); cam.m_Forward_dir = np.array(fwd, np.float32)
    r = renderer(env)
    r.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=depth, max_samples=frames + 1)
    r.ResizeBuffer(W, H)
    best = 1e9
    for _ in range(reps):
        r.resetAccumulationBuffer()
        t0 = time.perf_counter()
        r.RenderBatch(cam, sc, frames)
        best = min(best, time.perf_counter() - t0)
    img = r.GetRenderTargetImage()
    msg = "%-14s %4dx%-4d x%-2d d%-2d %-60s %8.3f ms %8.1f Msamples/s span %.3f ms" % (
        name, W, H, frames, depth, r.kernelInfo(), best * 1e3, W * H * frames / best / 1e6, r.kernelSpanMs())
    if check:
        osc = oracle.Scene.load_glb(scene_path(name)).build_bvh(20, 8)
        ocam = oracle.default_camera(position=pos, forward=fwd)
        ref, _, _ = oracle.render(osc, ocam, oracle.default_settings(ray_bounce_limit=depth), W, H, 1, frames)
        nbad = int((bits(img) != bits(ref)).any(axis=-1).sum())
        msg += "  | pixels differing from the oracle: %d" % nbad
    print(msg, flush=True)
    return img


if __name__ == "__main__":
    pool = {"DRT_KERNEL": "path_pool"}
    for k, v in os.environ.items():
        if k.startswith("DRT_POOL_"): pool[k] = v
    small = [("cornell_box", 64, 64, 1, 4), ("cornell_box", 160, 90, 3, 8), ("room", 128, 72, 2, 16), ("cornell_box", 256, 256, 1, 4),
             ("bvh_split_test", 96, 64, 2, 4), ("multi_material", 96, 64, 2, 4), ("cube_gltf", 64, 64, 2, 4)]
    for case in small:
        if case[0] in SCENES:
            run(*case, env=pool)
    for case in [("cornell_box", 1920, 1080, 8, 8), ("room", 1920, 1080, 4, 16)]:
        a = run(*case, env=pool, check=False)
        b = run(*case, env={"DRT_KERNEL": "wave_queue"}, check=False)
        print("   path_pool == wave_queue bit for bit:", bool((bits(a) == bits(b)).all()), flush=True)
