"""Host-side: oracle (CPU restatement) throughput against thread count.  python tools/oracle_threads.py [scene ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from tests.scenes import SCENES, scene_path
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count(), flush=True)
for name in (sys.argv[1:] or ["cornell_box", "cs16_dust"]):
    _, pos, fwd, depth = SCENES[name]
    sc = oracle.Scene.load_glb(scene_path(name)).build_bvh(20, 8)
    for th in (1, 4, 16, 32, 64, 128, 256):
        t0 = time.time()
        oracle.render(sc, oracle.default_camera(position=pos, forward=fwd), oracle.default_settings(ray_bounce_limit=depth), 960, 540, 1, 1, threads=th)
        dt = time.time() - t0
        print("%-12s threads %3d  %.2f s  %.3f Msamples/s" % (name, th, dt, 960 * 540 / dt / 1e6), flush=True)
