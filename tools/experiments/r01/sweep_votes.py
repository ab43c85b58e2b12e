"""GPU box: sweep the wave_queue voting thresholds on a workload (each setting in a fresh process)."""
import os, subprocess, sys
combos = [(12, 36, 4), (12, 36, 2), (12, 36, 8), (8, 36, 4), (16, 36, 4), (12, 28, 4), (12, 32, 4), (12, 40, 4), (12, 44, 4), (8, 32, 4), (16, 40, 4), (10, 32, 3), (14, 40, 6), (20, 36, 4)]
for n, s, r in combos:
    env = dict(os.environ, DRT_VOTE_N=str(n), DRT_VOTE_S=str(s), DRT_VOTE_R=str(r))
    out = subprocess.run([sys.executable, "tools/phase_stats.py"] + sys.argv[1:], env=env, capture_output=True, text=True).stdout.strip().splitlines()
    print("N %2d S %2d R %2d | %s | %s" % (n, s, r, out[-1].split("ms")[-1].strip() if out else "?", out[-2] if len(out) > 1 else ""), flush=True)
