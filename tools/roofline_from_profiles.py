"""The roofline numbers of one workload, recomputed from rocprofv3 output alone.

    python tools/roofline_from_profiles.py <workload> <kernel_stats.csv> <pmc dir or csv> [...] [--out profiles/r02_roofline_<workload>.json]
    e.g. from the committed files:
    python tools/roofline_from_profiles.py cornell_box_1080p_8spp_d8 profiles/r02_bench_f1_kernel_stats_cornell_box_1080p_8spp_d8.csv \
           profiles/r02_pmc_sq1_cornell_box_1080p_8spp_d8.csv profiles/r02_pmc_sq2_cornell_box_1080p_8spp_d8.csv \
           profiles/r02_pmc_fetch_cornell_box_1080p_8spp_d8.csv profiles/r02_pmc_write_cornell_box_1080p_8spp_d8.csv

Inputs (all produced by tools/collect_profiles.sh on the GPU box, one `bench.py --frames-in-flight 1` run each):
  kernel_stats.csv   rocprofv3 --kernel-trace --stats: average duration of the tracing kernel, launched alone
  pmc dirs           rocprofv3 --pmc passes (counter_collection.csv): SQ instruction / lane counters, GRBM_GUI_ACTIVE,
                     FETCH_SIZE and WRITE_SIZE (separate passes, MI355X_MICROARCH.md "HBM")
Output: per-launch counter averages of the tracing kernel and the derived fractions

  valu_issue   (SQ_INSTS_VALU + SQ_INSTS_VALU_FMA_F32 + 3 * SQ_INSTS_VALU_TRANS_F32) issue slots of 2 SIMD cycles
               / (1024 SIMDs x 2.4 GHz / 2 x kernel time)
               Issue costs measured on gfx950 (tools/microbench/valu_issue.hip, wave64): add / mul / min / max / cndmask / integer
               2 cycles, fma / mad 4, rcp / sqrt / rsq 8.  Compares that write a lane mask (4 cycles) and conversions have no
               counter of their own and are priced at 2: the fraction is a LOWER bound on the pipe's occupancy.
  lane_utilisation   SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)  (rocprofv3's own VALUUtilization)
  hbm_measured_frac  (2 x FETCH_SIZE + WRITE_SIZE) / kernel time / 8 TB/s  (read side doubled: the gfx950 correction)
bench.py reads the JSON written here for the counter values and divides by the kernel time IT measures.
"""
import collections, csv, glob, json, os, re, sys

SIMDS, CLOCK_HZ, HBM_PEAK = 1024, 2.4e9, 8.0e12          # MI355X_MICROARCH.md "Chip-level parameters"
TRACE_MARKERS = ("path_pool_kernel", "wave_queue_kernel", "pixel_walk_kernel")


def is_trace(name):
    return any(m in name for m in TRACE_MARKERS)


def trace_id(name):
    m = re.search(r"(\w+_kernel<[^>]*>)", name)
    return m.group(1) if m else name


def kernel_time(stats_csv):
    """-> (name, calls, average ns) of the tracing kernel with the largest total time"""
    best = None
    for r in csv.DictReader(open(stats_csv)):
        if is_trace(r["Name"]) and (best is None or float(r["TotalDurationNs"]) > best[3]):
            best = (r["Name"], int(r["Calls"]), float(r["AverageNs"]), float(r["TotalDurationNs"]))
    resolve = None
    for r in csv.DictReader(open(stats_csv)):
        if "resolve_kernel" in r["Name"]:
            resolve = (int(r["Calls"]), float(r["AverageNs"]))
    return best, resolve


def counters(dirs, kernel_name):
    """Per-dispatch averages of every counter collected for the dispatches of `kernel_name` (and of the resolve kernel)."""
    out = {"trace": collections.defaultdict(float), "resolve": collections.defaultdict(float)}
    n = {"trace": collections.defaultdict(set), "resolve": collections.defaultdict(set)}
    files = []
    for d in dirs:         # a rocprofv3 output directory, or a counter CSV itself (profiles/<tag>_pmc_*_<workload>.csv)
        files += [d] if os.path.isfile(d) else glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                key = "trace" if is_trace(name) and trace_id(name) == trace_id(kernel_name) else ("resolve" if "resolve_kernel" in name else None)
                if key is None:
                    continue
                out[key][r["Counter_Name"]] += float(r["Counter_Value"])
                n[key][r["Counter_Name"]].add((f, r["Dispatch_Id"]))
    return {k: {c: v / len(n[k][c]) for c, v in out[k].items()} for k in out}


def derive(c, kernel_s):
    """Fractions from the per-launch counter averages `c` of the tracing kernel and its duration."""
    d = {}
    if c.get("SQ_INSTS_VALU"):
        slots = c["SQ_INSTS_VALU"] + c.get("SQ_INSTS_VALU_FMA_F32", 0.0) + 3.0 * c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
        d["valu_issue_slots_per_launch"] = slots
        d["valu_issue_achieved_gslots_s"] = slots / kernel_s / 1e9
        d["valu_issue_peak_gslots_s"] = SIMDS * CLOCK_HZ / 2 / 1e9
        d["valu_issue_frac"] = slots / kernel_s / (SIMDS * CLOCK_HZ / 2)
    if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"):
        d["lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
    if c.get("SQ_WAVE_CYCLES"):
        # where a resident wave's cycles go (MI355X_MICROARCH.md "SQ": WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, disjoint)
        w = c["SQ_WAVE_CYCLES"]
        d["wave_time"] = {k: round(c[n] / w, 4) for k, n in (("parked_at_waitcnt", "SQ_WAIT_ANY"), ("issue_stalled", "SQ_WAIT_INST_ANY"),
                                                            ("issuing_any", "SQ_ACTIVE_INST_ANY"), ("issuing_scalar", "SQ_ACTIVE_INST_SCA"),
                                                            ("issuing_lds", "SQ_ACTIVE_INST_LDS")) if c.get(n) is not None}
        if c.get("SQ_ACTIVE_INST_VALU"):
            d["wave_time"]["issuing_valu"] = round(c["SQ_ACTIVE_INST_VALU"] / w, 4)
    if c.get("TCC_REQ_sum"):
        # L2 (MI355X_MICROARCH.md "L2": ~34.5 TB/s aggregate; a request moves at most one 128-byte line)
        req, hit, miss = c["TCC_REQ_sum"], c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
        d["l2"] = {"requests_per_launch": int(req), "hit_rate": round(hit / max(hit + miss, 1.0), 4),
                   "requests_per_s": req / kernel_s, "bandwidth_frac_upper_bound": round(req * 128.0 / kernel_s / 34.5e12, 4),
                   "note": "requests x 128 B / kernel time / 34.5 TB/s: an UPPER bound of the L2 bandwidth in use (most requests of this kernel are 16- to 64-byte gathers)"}
    if c.get("SQ_INSTS_VMEM_RD") and c.get("SQ_INSTS_VALU"):
        d["valu_per_vector_memory_read"] = round(c["SQ_INSTS_VALU"] / c["SQ_INSTS_VMEM_RD"], 1)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_share_of_lds_cycles"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
    if c.get("GRBM_GUI_ACTIVE"):
        d["shader_clock_ghz_during_kernel"] = c["GRBM_GUI_ACTIVE"] / 8.0 / kernel_s / 1e9      # the counter sums the 8 XCDs
    return d


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--out")]
    out_path = None
    for i, a in enumerate(sys.argv):
        if a == "--out":
            out_path = sys.argv[i + 1]
    args = [a for a in args if a != out_path]
    workload, stats_csv, dirs = args[0], args[1], args[2:]
    (name, calls, avg_ns, _), resolve = kernel_time(stats_csv)
    c = counters(dirs, name)
    kernel_s = avg_ns * 1e-9
    short = re.search(r"(\w+_kernel<[^>]*>)", name)
    res = {"workload": workload, "kernel": short.group(1) if short else name,
           "kernel_ms_rocprof": avg_ns * 1e-6, "kernel_launches_in_trace": calls,
           "resolve_ms_rocprof": resolve[1] * 1e-6 if resolve else None,
           "per_launch": {k: v for k, v in sorted(c["trace"].items())},
           "per_launch_resolve": {k: v for k, v in sorted(c["resolve"].items())}}
    res.update(derive(c["trace"], kernel_s))
    fetch_kib = c["trace"].get("FETCH_SIZE"), c["resolve"].get("FETCH_SIZE", 0.0)
    write_kib = c["trace"].get("WRITE_SIZE"), c["resolve"].get("WRITE_SIZE", 0.0)
    if fetch_kib[0] is not None and write_kib[0] is not None:
        trace_bytes = (2.0 * fetch_kib[0] + write_kib[0]) * 1024.0
        resolve_bytes = (2.0 * fetch_kib[1] + write_kib[1]) * 1024.0
        res["hbm_bytes_per_launch_trace"] = int(trace_bytes)
        res["hbm_bytes_per_launch"] = int(trace_bytes + resolve_bytes)           # tracing + resolve kernel of one launch
        res["hbm_bytes_per_launch_raw"] = int((fetch_kib[0] + write_kib[0] + fetch_kib[1] + write_kib[1]) * 1024.0)
        res["hbm_measured_frac"] = trace_bytes / kernel_s / HBM_PEAK
    res["constants"] = {"simds": SIMDS, "clock_hz": CLOCK_HZ, "hbm_peak_bytes_s": HBM_PEAK,
                        "issue_cycles": {"simple": 2, "fma": 4, "trans": 8}}
    text = json.dumps(res, indent=1)
    print(text)
    if out_path:
        open(out_path, "w").write(text + "\n")


if __name__ == "__main__":
    main()
