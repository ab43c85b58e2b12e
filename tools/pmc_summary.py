"""Summarise a rocprofv3 --pmc counter_collection.csv: per-kernel, per-dispatch averages."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "drt::" not in name:
                continue
            key = name.split("(")[0][-60:]
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[key].add(r["Dispatch_Id"])
        for key, d_ in agg.items():
            n = len(disp[key])
            print(f, key, "dispatches", n)
            for c, v in sorted(d_.items()):
                print("   %-26s %.5g" % (c, v / n))
            g = lambda c: d_.get(c, 0.0) / n
            if g("SQ_ACTIVE_INST_VALU"):
                print("   lane utilisation of VALU instructions = %.3f" % (g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64)))
                if g("SQ_WAVE_CYCLES"):
                    print("   VALU-active share of wave lifetime     = %.3f" % (g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")))
