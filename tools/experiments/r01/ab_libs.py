"""GPU box: A/B several builds of libdrt_hip.so on one workload, interleaved rounds in fresh processes.
   python tools/ab_libs.py libA.so libB.so ... -- scene W H spp"""
import os, subprocess, sys, statistics
sep = sys.argv.index("--")
libs = sys.argv[1:sep]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ, DRT_LIB_OVERRIDE=os.path.join(root, "dustraytracer_amd", l))
        out = subprocess.run([sys.executable, "tools/time_workload.py"] + sys.argv[sep + 1:], env=env, capture_output=True, text=True).stdout.strip().splitlines()
        res[l].append(float(out[-1].split(" ms ")[1].split()[0]))
for l in libs:
    print("%-24s ms median %.3f  min %.3f  all %s" % (l, statistics.median(res[l]), min(res[l]), res[l]))
