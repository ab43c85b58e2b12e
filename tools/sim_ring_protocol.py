#!/usr/bin/env python3
"""Exhaustive interleaving check of path_pool's LDS ring protocol (kernel_path_pool.hip: push_group / the claim).

A ring of `cap` 16-bit slots holds path ids; `tail` is advanced by producers (atomic add), `head` by consumers
(compare-and-swap).  Every wave alternates  claim one position -> read its slot -> clear it -> reserve a position ->
wait for that slot to be empty -> write the id back.  Each of those is ONE atomic LDS operation in the kernel and one
step here; the checker explores EVERY interleaving of the waves' steps (the whole global state graph, breadth first, so a
schedule it prints is a shortest one) and looks for a state in which a path id is in two hands (taken by two consumers)
-- the failure VERDICT r2 / ADVICE r2 describe: "a consumer of lap k+1 cannot tell a stale id of lap k from a fresh one".

  protocol "r02":  a consumer accepts any non-empty slot                      (round 2: hazard expected)
  protocol "lap":  a slot entry carries (position // cap) & 15 and a consumer accepts only the lap of its own position
                   (round 3, the shipped kernel: no hazard may exist)

Usage: python tools/sim_ring_protocol.py            # both protocols, cap = P = 2, three waves, three cycles each
Exit code 0 iff r02 shows the hazard (the model can see it) AND lap is clean.
"""
import collections
import sys

EMPTY = None


def explore(protocol, cap=2, n_ids=2, waves=3, cycles=3, max_states=5_000_000):
    """Returns (violating trace or None, states visited)."""
    # wave state: (pc, pos, id, cycles_left); pc in IDLE, CAS(h), READ(pos), CLEAR(pos,id), RESERVE(id), WAIT(pos,id)
    slots0 = tuple((i, 0) for i in range(n_ids)) + (EMPTY,) * (cap - n_ids)
    init = (0, n_ids, slots0, tuple(("IDLE", -1, -1, cycles) for _ in range(waves)))
    seen = {init}
    stack = collections.deque([(init, ())])            # breadth first: the first hazard found has the shortest schedule
    while stack:
        state, trace = stack.popleft()
        head, tail, slots, ws = state
        for w, (pc, pos, pid, left) in enumerate(ws):
            nh, nt, ns, nw = head, tail, slots, None
            if pc == "IDLE":
                if left == 0 or tail - head < 1:
                    continue                                  # nothing to claim (or this wave is done): no step
                nw = ("CAS", head, -1, left)                  # control read: remembers the head it saw
            elif pc == "CAS":
                if head == pos:
                    nh = head + 1
                    nw = ("READ", pos, -1, left)
                else:
                    nw = ("IDLE", -1, -1, left)               # lost the race
            elif pc == "READ":
                e = slots[pos % cap]
                ok = e is not EMPTY and (protocol == "r02" or e[1] == (pos // cap) % 16)
                if not ok:
                    continue                                  # spins: no state change
                # the id is now in this wave's hands: is it in anybody else's?
                for v, (pc2, _, pid2, _) in enumerate(ws):
                    if v != w and pc2 in ("CLEAR", "RESERVE", "WAIT") and pid2 == e[0]:
                        return trace + ((w, "READ pos %d takes id %d (entry of lap %d) -- already held by wave %d" % (pos, e[0], e[1], v)),), len(seen)
                nw = ("CLEAR", pos, e[0], left)
            elif pc == "CLEAR":
                ns = slots[:pos % cap] + (EMPTY,) + slots[pos % cap + 1:]
                nw = ("RESERVE", -1, pid, left)
            elif pc == "RESERVE":
                nt = tail + 1
                nw = ("WAIT", tail, pid, left)
            elif pc == "WAIT":
                if slots[pos % cap] is not EMPTY:
                    continue                                  # spins until the slot is empty
                ns = slots[:pos % cap] + ((pid, (pos // cap) % 16),) + slots[pos % cap + 1:]
                nw = ("IDLE", -1, -1, left - 1)
            nxt = (nh, nt, ns, ws[:w] + (nw,) + ws[w + 1:])
            if nxt in seen:
                continue
            seen.add(nxt)
            if len(seen) > max_states:
                raise SystemExit("state space larger than %d" % max_states)
            stack.append((nxt, trace + ((w, "%s pos=%d id=%d" % (pc, pos, pid)),)))
    return None, len(seen)


def main():
    ok = True
    for protocol in ("r02", "lap"):
        trace, n = explore(protocol)
        if trace is None:
            print("protocol %-3s: %d states, every interleaving explored, no path id ever in two hands" % (protocol, n))
        else:
            print("protocol %-3s: HAZARD after %d states; schedule (wave: step):" % (protocol, n))
            for w, what in trace:
                print("    wave %d: %s" % (w, what))
        ok = ok and ((trace is not None) == (protocol == "r02"))
    # a larger instance of the shipped protocol: ring of 4, 3 ids, 3 waves
    trace, n = explore("lap", cap=4, n_ids=3, waves=3, cycles=3)
    print("protocol lap, cap 4 / 3 ids / 3 waves: %d states, %s" % (n, "clean" if trace is None else "HAZARD"))
    ok = ok and trace is None
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
