# GPU box: bench lines of both tracing kernels on the judged workloads and the small-launch cases
# usage: bash tools/ab_kernels.sh OUTDIR
O=$1; mkdir -p $O
for k in path_pool wave_queue; do
  DRT_KERNEL=$k python3 bench.py --cpu-seconds 0 --no-roofline-counters > $O/c2_f3_$k.json 2>/dev/null
  DRT_KERNEL=$k python3 bench.py --cpu-seconds 0 --no-roofline-counters --frames-in-flight 1 > $O/c2_f1_$k.json 2>/dev/null
  DRT_KERNEL=$k python3 bench.py --cpu-seconds 0 --no-roofline-counters --workload cornell_box_256_1spp_d4 --steps 200 --warmup 20 > $O/c1_$k.json 2>/dev/null
  DRT_KERNEL=$k python3 bench.py --cpu-seconds 0 --no-roofline-counters --emulate-shard 0/8 --steps 200 --warmup 20 > $O/shard8_$k.json 2>/dev/null
  DRT_KERNEL=$k python3 bench.py --cpu-seconds 0 --no-roofline-counters --emulate-shard 0/8 --frames-in-flight 1 --steps 200 --warmup 20 > $O/shard8_f1_$k.json 2>/dev/null
  DRT_KERNEL=$k python3 bench.py --cpu-seconds 0 --no-roofline-counters --workload room_4k_64spp_d16 --steps 2 --warmup 1 > $O/c5_$k.json 2>/dev/null
done
python3 - $O <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-28s %10.1f Msamples/s  %9.4f ms/step" % (os.path.basename(f), d["value"], d["ms_per_step"]))
    except Exception as e:
        print(f, "FAILED", e)
PY
