# GPU box: how many levels of the traversal stacks to keep in LDS (DRT_POOL_STACK_LDS; the rest in HBM), per workload.
#   bash tools/experiments/r03/sweep_stack_lds.sh OUTDIR "workload ..." "k ..."
O=$1; mkdir -p $O
WLS=${2:-"room_4k_64spp_d16 dense_monkey_1080p_16spp_d2 suzanne_plane_1080p_8spp_d2 cs16_dust_1080p_8spp_d5 cornell_box_1080p_8spp_d8"}
KS=${3:-"99 2 3 4 5 6 8"}
for wl in $WLS; do
  for k in $KS; do
    steps=""; [ $wl = room_4k_64spp_d16 ] && steps="--steps 2 --warmup 1"
    DRT_POOL_STACK_LDS=$k python3 bench.py --cpu-seconds 0 --no-roofline-counters --workload $wl $steps > $O/${wl%%_*}_k$k.json 2>>$O/err.log || echo "FAILED $wl $k"
  done
done
python3 - $O <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print("%-28s %10.1f Msamples/s  %9.4f ms/step  kernel alone %s ms  %s" % (os.path.basename(f), d["value"], d["ms_per_step"], r.get("kernel_ms"), r.get("kernel")))
    except Exception as e:
        print(f, "FAILED", e)
PY
