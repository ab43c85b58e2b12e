"""Copy what tools/collect_profiles.sh left in gpurun_out/prof/ into profiles/ (files already carry the round's tag).
   python tools/install_profiles.py r02"""
import glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "prof"), os.path.join(root, "profiles")
n = 0
for f in sorted(glob.glob(os.path.join(src, tag + "_*"))):
    if os.path.getsize(f) > 3 << 20:
        print("skipped (too big)", f)
        continue
    shutil.copy(f, os.path.join(dst, os.path.basename(f)))
    n += 1
rows = []
for f in sorted(glob.glob(os.path.join(src, tag + "_bench_*.json"))):
    try:
        rows.append(json.loads(open(f).read().strip().splitlines()[-1]))
    except (ValueError, IndexError):
        print("no bench line in", f)
json.dump(rows, open(os.path.join(dst, "%s_bench_all_workloads.json" % tag), "w"), indent=1)
print("installed", n, "files,", len(rows), "bench lines")
