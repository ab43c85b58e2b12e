"""GPU box: time a workload under several env-var settings: python tools/sweep_env.py 'A=1,B=2' 'A=3,B=4' -- scene W H spp"""
import os, subprocess, sys
sep = sys.argv.index("--")
for combo in sys.argv[1:sep]:
    env = dict(os.environ)
    for kv in combo.split(","):
        if kv:
            k, v = kv.split("="); env[k] = v
    out = subprocess.run([sys.executable, "tools/time_workload.py"] + sys.argv[sep + 1:], env=env, capture_output=True, text=True).stdout.strip().splitlines()
    print("%-40s | %s" % (combo, out[-1].split("wave_queue")[-1] if out else "?"), flush=True)
