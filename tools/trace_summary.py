"""Summarise a rocprofv3 --kernel-trace csv: per (kernel, grid size) count / median / min duration in us."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        acc = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            name = "wave_queue" if "wave_queue" in name else "resolve" if "resolve" in name else name[:40]
            key = (name, int(r.get("Grid_Size_X", r.get("Grid_Size", 0))))
            acc.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in acc.items():
            v = sorted(v)
            print("%-12s grid %8d  n %3d  median %9.1f us  min %9.1f us" % (k[0], k[1], len(v), v[len(v) // 2], v[0]))
