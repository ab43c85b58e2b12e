"""Offline study of wave scheduling policies on real per-lane step traces from the oracle.

Each lane executes its own fixed sequence of steps (N = stack pop, T = triangle test, R = sampler try,
S = shade); a policy decides which phase the wave runs next; a phase run costs C[phase] issue slots
whatever the number of lanes it serves.  Output: issue slots per pixel for each policy / parameter set.
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from tests.scenes import SCENES, scene_path

COST = {"T": 85, "N": 100, "S": 600, "R": 75, "D": 130}


def get_traces(name="cornell_box", W=1920, H=1080, x0=900, y0=500, n=48, frames=8, depth=None):
    _, pos, fwd, d0 = SCENES[name]
    depth = depth or d0
    sc = oracle.Scene.load_glb(scene_path(name)).build_bvh(20, 8)
    cam = oracle.default_camera(position=pos, forward=fwd)
    st = oracle.default_settings(ray_bounce_limit=depth)
    buf = np.zeros(64 << 20, np.uint8)
    L = oracle.lib()
    L.o_trace_steps.restype = C.c_size_t
    cs = sc.c_scene()
    nbytes = L.o_trace_steps(C.byref(cs), C.byref(cam), C.byref(st), W, H, x0, y0, x0 + n, y0 + n, frames,
                             buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
    raw = buf[:nbytes].tobytes().decode()
    pixels = [p for p in raw.split("X") if p]
    return pixels


def lane_program(pixel_trace, merge_r):
    """Sequence of phase letters one lane executes for one pixel.  The S step of a ray is followed by its R tries
    (merge_r: the tries are part of S and cost extra issue slots there)."""
    prog = []
    for path in pixel_trace.split("P"):
        if not path:
            continue
        rays = path.split("S")          # each ray: N/T steps, then (after S) R tries of the NEXT ray's direction
        # trace order per ray: [R...] N T ... S ; R's come before the next ray's N
        for seg in path.replace("S", "S|").split("|"):
            if not seg:
                continue
            r = seg.count("R")
            body = seg.replace("R", "")
            if r:
                prog.extend(["R"] * r if not merge_r else [("r", r)])
            prog.extend(list(body))
    return prog


def simulate(pixels, policy, lanes=64, merge_r=True, thresholds=(20, 24, 24), paths_per_lane=1):
    """policy: 'vote' (thresholds θN, θS, θR) ; returns (slots, useful lane-slots)."""
    progs = [lane_program(p, merge_r) for p in pixels]
    queue = list(range(len(progs)))
    slots = np.zeros(1)
    P = lanes * paths_per_lane
    cur = [None] * P
    pos = [0] * P
    total = 0.0
    execs = {"T": 0, "N": 0, "S": 0, "R": 0}
    served = {"T": 0, "N": 0, "S": 0, "R": 0}

    def refill(i):
        if queue:
            cur[i] = progs[queue.pop()]
            pos[i] = 0
        else:
            cur[i] = None

    for i in range(P):
        refill(i)
    thN, thS, thR = thresholds
    while True:
        state = []
        for i in range(P):
            while cur[i] is not None and pos[i] >= len(cur[i]):
                refill(i)
            state.append(None if cur[i] is None else cur[i][pos[i]])
        cnt = {"T": 0, "N": 0, "S": 0, "R": 0}
        for s in state:
            if s is None:
                continue
            k = "S" if isinstance(s, tuple) else s
            cnt[k] += 1
        if sum(cnt.values()) == 0:
            break
        cap = lanes
        if cnt["S"] >= thS:
            ph = "S"
        elif cnt["R"] >= thR:
            ph = "R"
        elif cnt["N"] >= thN:
            ph = "N"
        elif cnt["T"] > 0:
            ph = "T"
        else:
            ph = max(("N", "S", "R"), key=lambda k: cnt[k])
        # serve up to `cap` lanes in that state
        n_served = 0
        rmax = 0
        for i in range(P):
            s = state[i]
            if s is None or n_served >= cap:
                continue
            k = "S" if isinstance(s, tuple) else s
            if k != ph:
                continue
            if isinstance(s, tuple):
                rmax = max(rmax, s[1])
            pos[i] += 1
            n_served += 1
        cost = COST[ph]
        if ph == "S" and merge_r:
            cost += rmax * COST["R"]
        if paths_per_lane > 1:
            cost += 25          # LDS state load/store per phase run
        total += cost
        execs[ph] += 1
        served[ph] += n_served
    return total, execs, served


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
    pixels = get_traces(name)
    npx = len(pixels)
    ideal = 0
    for p in pixels:
        ideal += p.count("T") * COST["T"] + p.count("N") * COST["N"] + p.count("S") * COST["S"] + p.count("R") * COST["R"]
    print(name, npx, "pixels; steps per pixel: T %.1f N %.1f S %.1f R %.1f; ideal lane-slots/pixel %.0f" % (
        sum(p.count("T") for p in pixels) / npx, sum(p.count("N") for p in pixels) / npx,
        sum(p.count("S") for p in pixels) / npx, sum(p.count("R") for p in pixels) / npx, ideal / npx))
    for label, kw in [("vote 3-class (R in S) 20/24", dict(merge_r=True, thresholds=(20, 24, 99))),
                      ("vote 3-class 12/16", dict(merge_r=True, thresholds=(12, 16, 99))),
                      ("vote 3-class 28/32", dict(merge_r=True, thresholds=(28, 32, 99))),
                      ("vote 4-class 20/24/24", dict(merge_r=False, thresholds=(20, 24, 24))),
                      ("vote 4-class 16/16/16", dict(merge_r=False, thresholds=(16, 16, 16))),
                      ("vote 4-class 12/12/12", dict(merge_r=False, thresholds=(12, 12, 12))),
                      ("pool x2 4-class 40/40/40", dict(merge_r=False, thresholds=(40, 40, 40), paths_per_lane=2)),
                      ("pool x2 4-class 56/56/56", dict(merge_r=False, thresholds=(56, 56, 56), paths_per_lane=2)),
                      ("pool x3 4-class 60/60/60", dict(merge_r=False, thresholds=(60, 60, 60), paths_per_lane=3)),
                      ("pool x4 4-class 64/64/64", dict(merge_r=False, thresholds=(64, 64, 64), paths_per_lane=4))]:
        total, execs, served = simulate(pixels, "vote", **kw)
        print("%-32s wave-slots/pixel %.0f  efficiency %.2f  | " % (label, total / npx, ideal / 64 / total) +
              "  ".join("%s %d@%.1f" % (k, execs[k], served[k] / max(execs[k], 1)) for k in "TNSR"))


def grid():
    pixels = get_traces("cornell_box", n=32)
    npx = len(pixels)
    ideal = sum(p.count("T") * COST["T"] + p.count("N") * COST["N"] + p.count("S") * COST["S"] + p.count("R") * COST["R"] for p in pixels)
    best = []
    for thN in (8, 12, 16, 20):
        for thS in (8, 12, 16, 20, 28):
            for thR in (8, 12, 16, 20, 28):
                total, execs, served = simulate(pixels, "vote", merge_r=False, thresholds=(thN, thS, thR))
                best.append((total / npx, thN, thS, thR, served["T"] / execs["T"]))
    best.sort()
    for b in best[:12]:
        print("slots/pixel %.0f  thN %d thS %d thR %d  T lanes %.1f  eff %.2f" % (b[0], b[1], b[2], b[3], b[4], ideal / 64 / npx / b[0]))


def simulate_own2(pixels, thresholds=(12, 36, 4), swap_cost=35, lanes=64, k_paths=2):
    """Each lane owns k_paths paths; in a phase run a lane serves ONE of its paths that is in that phase
    (register-resident alternative: the inactive path is swapped in with swap_cost extra issue slots per run that swaps)."""
    progs = [lane_program(p, False) for p in pixels]
    queue = list(range(len(progs)))
    cur = [[None] * k_paths for _ in range(lanes)]
    pos = [[0] * k_paths for _ in range(lanes)]
    active = [0] * lanes

    def refill(l, k):
        if queue:
            cur[l][k] = progs[queue.pop()]; pos[l][k] = 0
        else:
            cur[l][k] = None

    for l in range(lanes):
        for k in range(k_paths):
            refill(l, k)
    thN, thS, thR = thresholds
    total = 0.0
    execs = {"T": 0, "N": 0, "S": 0, "R": 0}; served = dict(execs)
    while True:
        st = []
        for l in range(lanes):
            row = []
            for k in range(k_paths):
                while cur[l][k] is not None and pos[l][k] >= len(cur[l][k]):
                    refill(l, k)
                row.append(None if cur[l][k] is None else cur[l][k][pos[l][k]])
            st.append(row)
        cnt = {"T": 0, "N": 0, "S": 0, "R": 0}
        for row in st:
            for ph in cnt:
                if ph in row:
                    cnt[ph] += 1            # lanes that COULD serve this phase with one of their paths
        if sum(cnt.values()) == 0:
            break
        if cnt["S"] >= thS: ph = "S"
        elif cnt["R"] >= thR: ph = "R"
        elif cnt["N"] >= thN: ph = "N"
        elif cnt["T"] > 0: ph = "T"
        else: ph = max(("N", "S", "R"), key=lambda k: cnt[k])
        n = 0; swapped = False
        for l in range(lanes):
            row = st[l]
            if ph not in row:
                continue
            k = active[l] if row[active[l]] == ph else row.index(ph)
            if k != active[l]:
                swapped = True; active[l] = k
            pos[l][k] += 1; n += 1
        total += COST[ph] + (swap_cost if swapped else 0)
        execs[ph] += 1; served[ph] += n
    return total, execs, served


def simulate_policy(pixels, choose, lanes=64):
    """Generic policy: choose(cnt) -> phase letter given counts of lanes per class."""
    progs = [lane_program(p, False) for p in pixels]
    queue = list(range(len(progs)))
    cur = [None] * lanes; pos = [0] * lanes
    def refill(i):
        if queue: cur[i] = progs[queue.pop()]; pos[i] = 0
        else: cur[i] = None
    for i in range(lanes): refill(i)
    total = 0.0; execs = {"T": 0, "N": 0, "S": 0, "R": 0}; served = dict(execs)
    while True:
        state = []
        for i in range(lanes):
            while cur[i] is not None and pos[i] >= len(cur[i]): refill(i)
            state.append(None if cur[i] is None else cur[i][pos[i]])
        cnt = {"T": 0, "N": 0, "S": 0, "R": 0}
        for s_ in state:
            if s_ is not None: cnt[s_] += 1
        if sum(cnt.values()) == 0: break
        ph = choose(cnt)
        n = 0
        for i in range(lanes):
            if state[i] == ph: pos[i] += 1; n += 1
        total += COST[ph]; execs[ph] += 1; served[ph] += n
    return total, execs, served
