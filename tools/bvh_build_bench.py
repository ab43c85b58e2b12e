"""Host vs GPU BVH build time (same tree): shipped scenes and synthetic soups.  Run on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dustraytracer_amd as drt
from tests.scenes import scene_path


def soup(n, seed=1):
    rng = np.random.default_rng(seed)
    pos = (rng.uniform(-10, 10, (n, 1, 3)) + rng.uniform(-0.05, 0.05, (n, 3, 3))).astype(np.float32)
    def make():
        sc = drt.Scene(); sc.addMaterial((0.8, 0.8, 0.8), -1)
        sc.setGeometry(pos, np.tile(np.array([0, 1, 0], np.float32), (n, 3, 1)), np.zeros((n, 3, 2), np.float32), np.zeros(n, np.int32))
        return sc
    return make


def from_file(name):
    def make():
        sc = drt.Scene(); sc.loadGLTFmodel(scene_path(name)); return sc
    return make


cases = [(n, from_file(n)) for n in ("cornell_box", "dense_monkey", "cs16_dust")] + [("soup_%dk" % (n // 1000), soup(n)) for n in (100_000, 1_000_000, 4_000_000)]
b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8
b.m_BuildDevice = 0; b.buildIterative(from_file("cornell_box")())       # warm-up: context, code object
print("%-14s %9s %7s %6s | %10s %12s %12s | same" % ("scene", "tris", "nodes", "depth", "host ms", "gpu ms(dev)", "gpu ms(call)"))
for name, make in cases:
    h = make(); b.m_BuildDevice = -1
    t0 = time.perf_counter(); b.buildIterative(h); t_host = (time.perf_counter() - t0) * 1e3
    d = make(); b.m_BuildDevice = 0
    t0 = time.perf_counter(); b.buildIterative(d); t_call = (time.perf_counter() - t0) * 1e3
    hn, dn = h.m_BVHNodes, d.m_BVHNodes
    same = len(hn) == len(dn) and all(np.array_equal(hn[f], dn[f]) for f in hn.dtype.names) and h.m_PrimitivesBuffer.tobytes() == d.m_PrimitivesBuffer.tobytes()
    print("%-14s %9d %7d %6d | %10.2f %12.2f %12.2f | %s" % (name, len(h.m_PrimitivesBuffer), len(hn), h.bvh_depth, t_host, b.m_LastBuildDeviceMs, t_call, same), flush=True)
