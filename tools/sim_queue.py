"""Offline model of the "path pool" kernel: path state parked in LDS, phase-homogeneous batches of 64.

A pool of P paths lives in LDS (per workgroup, shared by W waves; or per wave).  Every path waits in exactly one queue:
  N  (pop a stack entry / test both children)          T  (test the triangles of one leaf, two per step)
  H  (shade a hit)    R (bounce-direction tries)        L  (launch the bounce ray)     E  (finish path + new sample + primary ray)
A wave takes up to 64 paths of ONE queue, runs that phase for them and routes each path to its next queue.
Cost = VALU wave-instructions of a batch (incl. state load / store / routing).  Event driven: every wave has its own clock;
a batch's results become visible when it ends.  Reports wave-instructions per sample and batch fill, to be compared with the
production kernel's measured ~203 per sample on cornell_box (profiles/r01_wave_queue_pmc_sq.txt).
"""
import heapq, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sim_pool import get_paths

OVH = 40            # state load + store + routing of one batch
COST = {"N": 75 + OVH, "Tstep": 125, "T0": OVH + 5, "H": 200 + OVH, "R": 62 + 25, "L": 80 + OVH, "E": 330 + OVH}


def simulate(chunks, P=1024, W=8, shared=True, min_fill=48, t_refill=True, policy="fullest", verbose=False, t_classes=(2, 6, 8, 10)):
    order = [p for ch in chunks for p in ch]
    qi = [0]
    n_pools = 1 if shared else W
    # path record: [rays, ri, vi, rtries_left]
    tq = ["T%d" % i for i in range(len(t_classes))]
    qnames = ["N", "H", "R", "L", "E"] + tq
    queues = [{k: [] for k in qnames} for _ in range(n_pools)]
    for pool in range(n_pools):
        for _ in range(P if shared else P // W):
            queues[pool]["E"].append(None)          # empty slot: E deals it a sample
    clock = [0.0] * W
    busy = [0.0] * W
    idle = [0.0] * W
    batches = {k: [0, 0, 0.0] for k in qnames}   # runs, lanes, cost
    events = [(0.0, w) for w in range(W)]
    heapq.heapify(events)
    pending = []        # (time, pool, queue, path)
    live = [0]
    done_samples = [0]

    def route(p):
        rays, ri, vi = p[0], p[1], p[2]
        visits = rays[ri][0]
        if vi < len(visits):
            if visits[vi] == 0:
                return "N"
            st = (visits[vi] + 1) // 2
            for i, c in enumerate(t_classes):
                if st <= c:
                    return tq[i]
            return tq[-1]
        return "H" if ri + 1 < len(rays) else "E"

    while events:
        now, w = heapq.heappop(events)
        pool = 0 if shared else w
        # deliver finished batches
        keep = []
        for ev in pending:
            if ev[0] <= now and ev[1] == pool:
                queues[pool][ev[2]].append(ev[3])
            else:
                keep.append(ev)
        pending[:] = keep
        q = queues[pool]
        exhausted = qi[0] >= len(order)
        sizes = {k: len(q[k]) for k in q}
        if exhausted and sizes["E"] and all(x is None for x in q["E"]):
            sizes["E"] = 0
        cands = [k for k in sizes if sizes[k] > 0]
        if not cands:
            mine = [ev for ev in pending if ev[1] == pool]
            if not mine:
                continue                      # pool drained: wave exits
            t_next = min(ev[0] for ev in mine)
            idle[w] += t_next - now
            heapq.heappush(events, (t_next, w))
            continue
        full = [k for k in cands if sizes[k] >= 64]
        if policy == "fullest":
            k = max(full or cands, key=lambda c: sizes[c])
        else:   # prefer work that frees paths: E last unless nothing else
            pref = tq + list("NHLRE")
            k = next((c for c in pref if c in full), None) or max(cands, key=lambda c: sizes[c])
        if sizes[k] < min_fill:
            mine = [ev for ev in pending if ev[1] == pool]
            if mine:                           # wait for company rather than run a thin batch
                t_next = min(ev[0] for ev in mine)
                if t_next > now:
                    idle[w] += t_next - now
                    heapq.heappush(events, (t_next, w))
                    continue
        items = q[k][:64]
        del q[k][:64]
        if k == "E" and exhausted:
            items = [it for it in items if it is not None]
            if not items:
                heapq.heappush(events, (now, w))
                continue
        n = len(items)
        if k[0] == "T":
            steps = [(it[0][it[1]][0][it[2]] + 1) // 2 for it in items]
            cost = COST["T0"] + COST["Tstep"] * max(steps)
            useful = sum(steps) / max(steps)
        elif k == "R":
            # loop with refill from the queue: a lane keeps an item for its number of tries; model: total tries / 64 steps
            tries = sum(it[3] for it in items)
            steps = max(max(it[3] for it in items), (tries + 63) // 64) if not t_refill else max(1, (tries + 63) // 64)
            cost = COST["R"] * steps
            useful = tries / steps
        else:
            cost = COST[k]
            useful = n
        batches[k][0] += 1; batches[k][1] += useful; batches[k][2] += cost
        end = now + cost
        busy[w] += cost
        for it in items:
            if k == "E":
                if it is not None:
                    done_samples[0] += 1
                if qi[0] < len(order):
                    p = [order[qi[0]], 0, 0, 0]; qi[0] += 1
                    pending.append((end, pool, route(p), p))
                # else the slot retires
            elif k[0] == "T" or k == "N":
                it[2] += 1
                pending.append((end, pool, route(it), it))
            elif k == "H":
                it[3] = max(1, it[0][it[1]][1])
                pending.append((end, pool, "R", it))
            elif k == "R":
                pending.append((end, pool, "L", it))
            elif k == "L":
                it[1] += 1; it[2] = 0
                pending.append((end, pool, route(it), it))
        heapq.heappush(events, (end, w))
    ns = len(order)
    total = sum(b[2] for b in batches.values())
    out = "instr/sample %6.1f  idle %4.1f%% | " % (total / ns, 100 * sum(idle) / max(sum(idle) + sum(busy), 1))
    out += "  ".join("%s %.2f@%.0f" % (k, batches[k][0] / ns * 64, batches[k][1] / max(batches[k][0], 1)) for k in qnames)
    return out


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
    depth = {"cornell_box": 8, "room": 16}.get(name)
    regs = [(900, 500, 32), (200, 100, 16), (1500, 800, 16), (960, 200, 16)]
    frames = 2 if name == "room" else 8
    chunks = get_paths(name, 1920, 1080, depth, regs, frames)
    print(name, len(chunks), "chunks")
    for label, kw in [("shared P=1024 W=8", dict(P=1024, W=8)),
                      ("shared P=1024 W=12", dict(P=1024, W=12)),
                      ("shared P=768 W=8", dict(P=768, W=8)),
                      ("shared P=1536 W=12", dict(P=1536, W=12)),
                      ("private P=8x128", dict(P=1024, W=8, shared=False, min_fill=0)),
                      ("private P=8x192", dict(P=1536, W=8, shared=False, min_fill=0)),
                      ("private P=8x256", dict(P=2048, W=8, shared=False, min_fill=0)),
                      ("shared P=1024 W=8 prio", dict(P=1024, W=8, policy="prio")),
                      ]:
        print("%-26s %s" % (label, simulate(chunks, **kw)))
