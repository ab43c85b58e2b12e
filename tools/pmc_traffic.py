"""HBM traffic of the tracing kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <workload> [out.json]

Method (MI355X_MICROARCH.md "HBM" + cdna_hip_programming.md 7): FETCH_SIZE / WRITE_SIZE are in KiB and come from the
L2's memory-side request counters, collected in SEPARATE passes (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2).
On gfx950 FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming read, so the read side is doubled;
WRITE_SIZE is exact for 16-byte-per-lane stores.  Our access mix (scattered 3-byte texel reads, 16-byte sample stores,
streaming resolve) is not a calibrated pattern, so both the raw and the corrected figures are kept.
Per launch = tracing kernel + resolve kernel of one bench step.
"""
import collections, csv, glob, json, sys


def per_dispatch(d, counter):
    tot = collections.defaultdict(float)
    n = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or "drt::" not in r["Kernel_Name"]:
                continue
            key = "resolve" if "resolve_kernel" in r["Kernel_Name"] else ("trace" if ("wave_queue" in r["Kernel_Name"] or "pixel_walk" in r["Kernel_Name"]) else None)
            if key is None:
                continue
            tot[key] += float(r["Counter_Value"])
            n[key].add(r["Dispatch_Id"])
    return {k: tot[k] / len(n[k]) for k in tot}


fetch = per_dispatch(sys.argv[1], "FETCH_SIZE")
write = per_dispatch(sys.argv[2], "WRITE_SIZE")
out = {"workload": sys.argv[3], "unit": "bytes per launch (tracing kernel + resolve kernel)",
       "fetch_size_kib_raw": fetch, "write_size_kib_raw": write}
raw = sum(fetch.values()) * 1024 + sum(write.values()) * 1024
corrected = 2 * sum(fetch.values()) * 1024 + sum(write.values()) * 1024
out["hbm_bytes_per_launch_raw"] = int(raw)
out["hbm_bytes_per_launch"] = int(corrected)
out["note"] = "read side doubled per the gfx950 FETCH_SIZE correction; access pattern not separately calibrated"
print(json.dumps(out, indent=1))
if len(sys.argv) > 4:
    json.dump(out, open(sys.argv[4], "w"), indent=1)
