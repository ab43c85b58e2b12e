"""GPU box: sweep path_pool launch parameters on one workload.  usage: pool_sweep.py scene W H frames depth  (env sets in the list below)"""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from tools.pool_check import run  # (tools/pool_check.py)

if __name__ == "__main__":
    name, W, H, frames, depth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    run(name, W, H, frames, depth, env={"DRT_KERNEL": "wave_queue"}, check=False)
    for spec in sys.argv[6:]:
        env = {"DRT_KERNEL": "path_pool"}
        for kv in spec.split(","):
            k, v = kv.split("=")
            env["DRT_POOL_" + k] = v
        print(spec, end="  ")
        run(name, W, H, frames, depth, env=env, check=False)
