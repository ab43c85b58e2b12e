"""Offline model of the tracing kernel's wave scheduling, replaying real per-lane step traces from the oracle.

Unlike sim_schedule.py this one models the production kernel's actual structure -- two triangles per T step, leaf
visits as (N, ceil(k/2) x T), one S run per ray when the bounce direction is ready, background R tries -- so that the
run counts can be checked against the measured ones (profiles/r01_phase_stats_cornell.txt) before a new design is
judged with it.  Designs:
   own K   : every lane owns K paths (state of the inactive ones parked in LDS or registers); a phase run serves every
             lane that has ANY path in that phase  (K = 1 is the production kernel)
Cost = VALU wave-instructions per phase run (+ per-run overhead for K > 1).
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from tests.scenes import SCENES, scene_path

COST = {"T": 150, "N": 100, "S": 500, "R": 60}


def get_paths(name, W, H, depth, regions, frames):
    """-> list of paths; a path = list of rays; a ray = (visits, r_tries) with visits = list of triangle counts per pop
    (0 = interior or culled pop), r_tries = sampler tries drawn after this ray's hit (0 = none)."""
    _, pos, fwd, d0 = SCENES[name]
    depth = depth or d0
    sc = oracle.Scene.load_glb(scene_path(name)).build_bvh(20, 8)
    cam = oracle.default_camera(position=pos, forward=fwd)
    st = oracle.default_settings(ray_bounce_limit=depth)
    buf = np.zeros(256 << 20, np.uint8)
    L = oracle.lib()
    L.o_trace_steps.restype = C.c_size_t
    cs = sc.c_scene()
    chunks = []
    for (x0, y0, n) in regions:
        nbytes = L.o_trace_steps(C.byref(cs), C.byref(cam), C.byref(st), W, H, x0, y0, x0 + n, y0 + n, frames,
                                 buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
        raw = buf[:nbytes].tobytes().decode()
        pixels = [p for p in raw.split("X") if p]
        assert len(pixels) == n * n
        # pixel-major, frames inside ('P' ends a path)
        per_pixel = [[q for q in p.split("P") if q] for p in pixels]
        for ty in range(n // 8):
            for tx in range(n // 8):
                for f in range(frames):
                    chunk = []
                    for k in range(64):
                        px = (ty * 8 + (k >> 3)) * n + tx * 8 + (k & 7)
                        chunk.append(parse_path(per_pixel[px][f]))
                    chunks.append(chunk)
    return chunks


def parse_path(s):
    rays = []
    for seg in s.split("S")[:-1] if s.endswith("S") else s.split("S"):
        r = 0
        i = 0
        while i < len(seg) and seg[i] == "R":
            r += 1; i += 1
        body = seg[i:]
        visits = []
        for ch in body:
            if ch == "N":
                visits.append(0)
            elif ch in "Tt":
                visits[-1] += 1
            elif ch == "R":
                r += 1
        if rays:
            rays[-1] = (rays[-1][0], r)      # tries drawn after the previous ray's hit
        elif r:
            pass
        rays.append((visits, 0))
    return rays


class Path:
    __slots__ = ("rays", "ri", "vi", "t_left", "cls", "pend", "blocked")


def simulate(chunks, K=1, vote_n=12, vote_s=44, vote_r_block=4, vote_spec=8, overhead=0, lanes=64, s_any=False, verbose=False):
    """Returns dict of runs and served lanes per phase, total cost."""
    order = [p for ch in chunks for p in ch]
    qi = [0]

    def new_path():
        if qi[0] >= len(order):
            return None
        p = Path(); p.rays = order[qi[0]]; qi[0] += 1
        p.ri = 0; p.vi = 0; p.t_left = 0; p.pend = 0; p.blocked = False
        start_ray(p)
        return p

    def start_ray(p):
        # S just launched ray ri: arm the background direction (tries known from the trace: r_tries of this ray)
        p.vi = 0; p.t_left = 0
        p.pend = p.rays[p.ri][1]
        p.blocked = False
        p.cls = "N" if p.rays[p.ri][0] else "S"

    slots = [[new_path() for _ in range(K)] for _ in range(lanes)]
    runs = {k: 0 for k in "TNSR"}; served = {k: 0 for k in "TNSR"}
    total = 0

    while True:
        cnt = {k: 0 for k in "TNSR"}
        n_pend = 0; alive = 0
        for l in range(lanes):
            has = set()
            for p in slots[l]:
                if p is None: continue
                alive += 1
                has.add("R" if (p.cls == "S" and p.blocked) else p.cls)
                if p.pend > 0: n_pend += 1
            for k in has: cnt[k] += 1
        if alive == 0:
            break
        exhausted = qi[0] >= len(order)
        vn = 4 if exhausted else vote_n
        vs = 36 if exhausted else vote_s
        # production order: S vote, R loop, N loop, T loop (one step each here; the loops emerge from repetition)
        if cnt["S"] >= vs or (cnt["T"] == 0 and cnt["N"] == 0 and cnt["S"] > 0 and cnt["S"] >= cnt["R"]):
            ph = "S"
        elif n_pend > 0 and (n_pend >= vote_spec or cnt["R"] >= vote_r_block or (cnt["R"] > 0 and cnt["T"] == 0 and cnt["N"] == 0)):
            ph = "R"
        elif cnt["N"] > 0 and (cnt["N"] >= vn or cnt["T"] == 0):
            ph = "N"
        elif cnt["T"] > 0:
            ph = "T"
        elif cnt["S"] > 0:
            ph = "S"
        else:
            ph = "R"
        runs[ph] += 1
        total += COST[ph] + (overhead if K > 1 else 0)
        if ph == "R":
            n = 0
            for l in range(lanes):
                did = False
                for p in slots[l]:
                    if p is not None and p.pend > 0 and not did:
                        p.pend -= 1; did = True
                        if p.pend == 0 and p.blocked: p.blocked = False
                if did: n += 1
            served["R"] += n
            continue
        n = 0
        for l in range(lanes):
            for k in range(K):
                p = slots[l][k]
                if p is None: continue
                c = "R" if (p.cls == "S" and p.blocked) else p.cls
                if c != ph: continue
                n += 1
                if ph == "T":
                    p.t_left -= 2
                    if p.t_left <= 0:
                        p.vi += 1
                        p.cls = "N" if p.vi < len(p.rays[p.ri][0]) else "S"
                elif ph == "N":
                    k_tris = p.rays[p.ri][0][p.vi]
                    if k_tris > 0:
                        p.t_left = k_tris; p.cls = "T"
                    else:
                        p.vi += 1
                        p.cls = "N" if p.vi < len(p.rays[p.ri][0]) else "S"
                else:   # S
                    last = p.ri + 1 >= len(p.rays)
                    if last:
                        slots[l][k] = new_path()      # store + deal + primary ray in the same run
                    elif p.pend > 0:
                        p.blocked = True              # shaded; direction not ready: wait for R, then another S run
                        p.cls = "S"
                        # mark shaded so the next S run only launches: model as same cost
                    else:
                        p.ri += 1
                        start_ray(p)
                break        # one path per lane per run
        served[ph] += n
    nsamp = len(order)
    out = {"cost_per_sample": total / nsamp}
    for k in "TNSR":
        out[k] = (runs[k] / nsamp * 64, served[k] / max(runs[k], 1))
    util = sum(served[k] * COST[k] for k in "TNSR") / (64.0 * total)
    out["util"] = util
    return out


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
    W, H = 1920, 1080
    depth = {"cornell_box": 8, "room": 16}.get(name)
    regs = [(900, 500, 32), (200, 100, 16), (1500, 800, 16), (960, 200, 16)]
    frames = 4 if name == "room" else 8
    chunks = get_paths(name, W, H, depth, regs, frames)
    print(name, len(chunks), "chunks")
    for label, kw in [("production K=1", dict(K=1)),
                      ("K=2 +0", dict(K=2, overhead=0)), ("K=2 +20", dict(K=2, overhead=20)), ("K=2 +40", dict(K=2, overhead=40)),
                      ("K=2 +20 votes 24/56", dict(K=2, overhead=20, vote_n=24, vote_s=56)),
                      ("K=3 +20 votes 32/60", dict(K=3, overhead=20, vote_n=32, vote_s=60)),
                      ("K=4 +20 votes 40/62", dict(K=4, overhead=20, vote_n=40, vote_s=62))]:
        o = simulate(chunks, **kw)
        print("%-26s cost/sample %7.0f util %.3f | " % (label, o["cost_per_sample"], o["util"]) +
              "  ".join("%s %.3f@%.1f" % (k, o[k][0], o[k][1]) for k in "TNSR"))
