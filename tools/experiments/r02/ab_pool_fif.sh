O=$1; mkdir -p $O
for sg in 1 0; do
  DRT_POOL_SHARE_GRID=$sg python3 bench.py --cpu-seconds 0 --no-roofline-counters > $O/c2_f3_sg$sg.json 2>/dev/null
  DRT_POOL_SHARE_GRID=$sg python3 bench.py --cpu-seconds 0 --no-roofline-counters --emulate-shard 0/8 --steps 200 --warmup 20 > $O/shard8_f3_sg$sg.json 2>/dev/null
  DRT_POOL_SHARE_GRID=$sg python3 bench.py --cpu-seconds 0 --no-roofline-counters --emulate-shard 0/8 --frames-in-flight 2 --steps 200 --warmup 20 > $O/shard8_f2_sg$sg.json 2>/dev/null
  DRT_POOL_SHARE_GRID=$sg python3 bench.py --cpu-seconds 0 --no-roofline-counters --frames-in-flight 2 > $O/c2_f2_sg$sg.json 2>/dev/null
done
python3 bench.py --cpu-seconds 0 --no-roofline-counters --workload room_4k_64spp_d16 --steps 2 --warmup 1 --frames-in-flight 1 > $O/c5_f1_pool.json 2>/dev/null
DRT_KERNEL=wave_queue python3 bench.py --cpu-seconds 0 --no-roofline-counters --workload room_4k_64spp_d16 --steps 2 --warmup 1 --frames-in-flight 1 > $O/c5_f1_wq.json 2>/dev/null
python3 tools/pool_stats.py room 1920 1080 4 16 > $O/stats_room.txt 2>&1
python3 - $O <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-28s %10.1f Msamples/s  %9.4f ms/step" % (os.path.basename(f), d["value"], d["ms_per_step"]))
    except Exception as e:
        print(f, "FAILED", e)
PY
cat $O/stats_room.txt
