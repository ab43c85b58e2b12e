"""Offline estimate: how many triangle tests per ray survive if every leaf's triangles are grouped in pairs and a pair is
skipped when the ray misses the pair's box or enters it beyond the closest hit.  Random diffuse rays from surfaces."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, oracle
from tests.scenes import scene_path
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
sc = oracle.Scene.load_glb(scene_path(name)).build_bvh(20, 8)
T = sc.tris; P = T['p'].astype(np.float64); N = len(T)
nodes = sc.nodes
leaves = [(int(n['prim_start']), int(n['prim_count']), n['bmin'].astype(float), n['bmax'].astype(float)) for n in nodes if n['is_leaf']]
rng = np.random.default_rng(0)
def greedy_pairs(idx):
    idx = list(idx); pairs = []
    def area(b0, b1): d = b1 - b0; return d[0]*d[1] + d[1]*d[2] + d[0]*d[2]
    bb = {i: (P[i].min(0), P[i].max(0)) for i in idx}
    while len(idx) > 1:
        best = None
        for a in range(len(idx)):
            for b in range(a + 1, len(idx)):
                i, j = idx[a], idx[b]
                ar = area(np.minimum(bb[i][0], bb[j][0]), np.maximum(bb[i][1], bb[j][1]))
                if best is None or ar < best[0]: best = (ar, i, j)
        _, i, j = best; pairs.append((i, j)); idx.remove(i); idx.remove(j)
    if idx: pairs.append((idx[0],))
    return pairs
def seq_pairs(idx):
    idx = list(idx); return [tuple(idx[k:k + 2]) for k in range(0, len(idx), 2)]
def boxes(pairs):
    return [(np.min([P[i].min(0) for i in p], 0), np.max([P[i].max(0) for i in p], 0)) for p in pairs]
def slab(o, inv, b0, b1):
    t0 = (b0 - o) * inv; t1 = (b1 - o) * inv
    tn = np.max(np.minimum(t0, t1)); tf = np.min(np.maximum(t0, t1))
    return max(tn, 0.0) if tf >= max(tn, 0) else -1.0
def mt(o, d, p):
    e1 = p[1] - p[0]; e2 = p[2] - p[0]; pv = np.cross(d, e2); det = e1 @ pv
    if abs(det) < 1e-12: return None
    tv = o - p[0]; u = (tv @ pv) / det
    if u < 0 or u > 1: return None
    q = np.cross(tv, e1); v = (d @ q) / det
    if v < 0 or u + v > 1: return None
    t = (e2 @ q) / det
    return t if t > 1e-6 else None
# rays: from a random point on a random triangle (area weighted), direction = normal + unit sphere point
area = 0.5 * np.linalg.norm(np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0]), axis=1)
cent = P.mean(1).mean(0)
stats = {"greedy": [0, 0, 0], "seq": [0, 0, 0]}
plan = {"greedy": [(greedy_pairs(range(s, s + c))) for s, c, _, _ in leaves], "seq": [seq_pairs(range(s, s + c)) for s, c, _, _ in leaves]}
pbox = {k: [boxes(pp) for pp in v] for k, v in plan.items()}
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
full = 0
for r in range(nr):
    i = rng.choice(N, p=area / area.sum()); a, b = rng.random(2)
    if a + b > 1: a, b = 1 - a, 1 - b
    o = P[i, 0] + a * (P[i, 1] - P[i, 0]) + b * (P[i, 2] - P[i, 0])
    n = np.cross(P[i, 1] - P[i, 0], P[i, 2] - P[i, 0]); n /= np.linalg.norm(n)
    if n @ (cent - o) < 0: n = -n
    while True:
        s = rng.normal(size=3); s /= np.linalg.norm(s); break
    d = n + s
    if np.linalg.norm(d) < 1e-3: continue
    o = o + n * 1e-3; inv = 1.0 / np.where(d == 0, 1e-30, d)
    # closest hit over everything (what hit_t converges to); leaves visited near-first approximated by box distance
    order = sorted(range(len(leaves)), key=lambda k: slab(o, inv, leaves[k][2], leaves[k][3]))
    for mode in ("greedy", "seq"):
        hit_t = np.inf; tests = 0; btests = 0
        for k in order:
            dl = slab(o, inv, leaves[k][2], leaves[k][3])
            if dl < 0 or dl >= hit_t: continue
            for pp, (b0, b1) in zip(plan[mode][k], pbox[mode][k]):
                btests += 1
                dd = slab(o, inv, b0 - 1e-3, b1 + 1e-3)
                if dd < 0 or dd * 0.99 >= hit_t: continue
                for j in pp:
                    tests += 1
                    t = mt(o, d, P[j])
                    if t is not None and t < hit_t: hit_t = t
        stats[mode][0] += tests; stats[mode][1] += btests
    hit_t = np.inf
    for k in order:
        dl = slab(o, inv, leaves[k][2], leaves[k][3])
        if dl < 0 or dl >= hit_t: continue
        for j in range(leaves[k][0], leaves[k][0] + leaves[k][1]):
            full += 1
            t = mt(o, d, P[j])
            if t is not None and t < hit_t: hit_t = t
print(name, "rays", nr, "plain tri tests/ray %.1f" % (full / nr))
for mode, (t, b, _) in stats.items():
    print("  %-6s pair-box tests/ray %.1f  tri tests/ray %.1f" % (mode, b / nr, t / nr))
