"""Copy what tools/collect_profiles.sh left in gpurun_out/prof/ into profiles/ under the round's names.
   python tools/install_profiles.py r01"""
import glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "prof"), os.path.join(root, "profiles")
plain = {"bench_default.json": "%s_bench_default.json", "bench_default_under_rocprof.json": "%s_bench_default_under_rocprof.json",
         "bench_default_f3_kernel_stats.csv": "%s_bench_default_f3_kernel_stats.csv",
         "bench_f1_under_rocprof.json": "%s_bench_f1_isolated.json", "bench_f1_isolated_kernel_stats.csv": "%s_bench_f1_isolated_kernel_stats.csv",
         "wave_queue_pmc_sq.txt": "%s_wave_queue_pmc_sq.txt", "phase_stats_cornell.txt": "%s_phase_stats_cornell.txt",
         "phase_stats_cs16_dust.txt": "%s_phase_stats_cs16_dust.txt",
         "bvh_build_host_vs_gpu.txt": "%s_bvh_build_host_vs_gpu.txt", "bvh_build_kernel_stats.csv": "%s_bvh_build_kernel_stats.csv"}
for a, b in plain.items():
    shutil.copy(os.path.join(src, a), os.path.join(dst, b % tag))
for f in glob.glob(os.path.join(src, "traffic_*.json")):
    shutil.copy(f, os.path.join(dst, os.path.basename(f)))              # bench.py reads these by workload name
for f in glob.glob(os.path.join(src, "pmc_*_size_*.csv")):
    shutil.copy(f, os.path.join(dst, "%s_%s" % (tag, os.path.basename(f))))
rows = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(os.path.join(src, "bench_*.json")))]
json.dump(rows, open(os.path.join(dst, "%s_bench_all_workloads.json" % tag), "w"), indent=1)
print("installed", len(plain) + len(rows), "files")
