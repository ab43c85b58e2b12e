"""GPU box: statistics of the path_pool kernel on a workload.  usage: pool_stats.py scene W H frames depth [K=V,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dustraytracer_amd as drt
from tests.scenes import SCENES, scene_path
from tools.pool_check import renderer

name, W, H, frames, depth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
env = {"DRT_KERNEL": "path_pool", "DRT_POOL_STATS": "1"}
for kv in (sys.argv[6].split(",") if len(sys.argv) > 6 else []):
    k, v = kv.split("="); env["DRT_POOL_" + k] = v
sc = drt.Scene(); sc.loadGLTFmodel(scene_path(name))
b = drt.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8; b.buildIterative(sc)
_, pos, fwd, _ = SCENES[name]
cam = drt.Camera(pos); cam.m_Forward_dir = np.array(fwd, np.float32)
r = renderer(env)
r.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=depth, max_samples=frames + 1)
r.ResizeBuffer(W, H)
r.RenderBatch(cam, sc, frames); r.poolStats()
r.resetAccumulationBuffer(); ms = r.RenderBatch(cam, sc, frames)
st = r.poolStats()
ns = W * H * frames
print(r.kernelInfo(), "%.3f ms (stats build)" % ms)
wt = st["wave_ticks"]
tot_b = 0
for q in ("N", "T0", "T1", "T2", "T3", "B", "E", "R", "S"):
    b_, l_, t_ = st[q]
    tot_b += b_
    print("  %-3s batches/64 samples %6.3f  fill %5.1f  ticks/batch %7.0f  share of wave time %5.1f%%" % (q, b_ / ns * 64, l_, t_ / max(b_, 1), 100.0 * t_ / wt))
print("  claims given up: %d (%.2f per batch), %.1f%% of wave time; waiting for company / idle: %.1f%% of wave time" % (
    st["failed_claims"], st["failed_claims"] / max(tot_b, 1), 100.0 * st["failed_claim_ticks"] / wt, 100.0 * st["idle_ticks"] / wt))
print("  claim: %.1f%% of wave time, %.0f ticks per batch; idle polls %d (%.2f per batch); lost claims %d" % (
    100.0 * st["claim_ticks"] / wt, st["claim_ticks"] / max(tot_b, 1), st["idle_polls"], st["idle_polls"] / max(tot_b, 1), st["lost_claims"]))
w = st.get("work")
if w:
    print("  lanes busy inside the loops: N %.1f of 64 over %.2f iterations per batch; T %.1f of 128 triangle slots per step, %.2f steps per batch; direction tries %.1f of 64 over %.2f iterations per B / R / S batch" % (
        w["n_lane_pops"] / max(w["n_iterations"], 1), w["n_iterations"] / max(st["N"][0], 1),
        w["t_lane_tests"] / max(w["t_steps"], 1), w["t_steps"] / max(sum(st[q][0] for q in ("T0", "T1", "T2", "T3")), 1),
        w["dir_lane_tries"] / max(w["dir_iterations"], 1), w["dir_iterations"] / max(st["B"][0] + st["R"][0] + st["S"][0], 1)))
