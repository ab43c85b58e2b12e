# GPU box: the judged bench lines for one or more builds of the library, side by side.
#   bash tools/experiments/r03/ab_quick.sh OUTDIR [lib.so ...]      (no lib = the in-tree libdrt_hip.so)
# Each line: C2 with 3 frames in flight (the bench default), C2 one frame at a time, C5 (room 4K / 64 spp), 1/8 shard of C2.
O=$1; shift; mkdir -p $O
LIBS="$@"; [ -z "$LIBS" ] && LIBS=dustraytracer_amd/libdrt_hip.so
for lib in $LIBS; do
  tag=$(basename $lib .so)
  export DRT_LIB_OVERRIDE=$PWD/$lib
  python3 bench.py --cpu-seconds 0 --no-roofline-counters > $O/c2_f3_$tag.json 2>$O/err_$tag.log || echo "FAILED c2_f3 $tag"
  python3 bench.py --cpu-seconds 0 --no-roofline-counters --frames-in-flight 1 --steps 30 > $O/c2_f1_$tag.json 2>>$O/err_$tag.log || echo "FAILED c2_f1 $tag"
  python3 bench.py --cpu-seconds 0 --no-roofline-counters --workload room_4k_64spp_d16 --steps 2 --warmup 1 > $O/c5_$tag.json 2>>$O/err_$tag.log || echo "FAILED c5 $tag"
  [ -n "$AB_SHARD" ] && python3 bench.py --cpu-seconds 0 --no-roofline-counters --emulate-shard 0/8 --steps 200 --warmup 20 > $O/shard8_$tag.json 2>>$O/err_$tag.log
  [ -n "$AB_MORE" ] && for wl in suzanne_plane_1080p_8spp_d2 dense_monkey_1080p_16spp_d2 cs16_dust_1080p_8spp_d5; do
    python3 bench.py --cpu-seconds 0 --no-roofline-counters --workload $wl > $O/${wl%%_*}_$tag.json 2>>$O/err_$tag.log || echo "FAILED $wl $tag"
  done
done
unset DRT_LIB_OVERRIDE
python3 - $O <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = (d.get("roofline") or {}).get("kernel_ms")
        print("%-36s %10.1f Msamples/s  %9.4f ms/step  kernel alone %s ms" % (os.path.basename(f), d["value"], d["ms_per_step"], k))
    except Exception as e:
        print(f, "FAILED", e)
PY
