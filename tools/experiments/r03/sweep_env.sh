# GPU box: bench lines of C2 (3 frames in flight) and C5 under a list of environment settings.
#   bash tools/experiments/r03/sweep_env.sh OUT "A=1 B=2" "A=2" ...
O=$1; shift; mkdir -p $O
i=0
for setting in "$@"; do
  i=$((i+1))
  c2=""; [ -z "$SWEEP_NO_C2" ] && c2=$(env $setting python3 bench.py --cpu-seconds 0 --no-roofline-counters --steps 40 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms/step (alone %.3f)' % (d['ms_per_step'], d['roofline']['kernel_ms']))")
  c5=""
  [ -z "$SWEEP_NO_C5" ] && c5=$(env $setting python3 bench.py --cpu-seconds 0 --no-roofline-counters --workload room_4k_64spp_d16 --steps 2 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f ms/step (alone %.2f)' % (d['ms_per_step'], d['roofline']['kernel_ms']))")
  echo "$setting | C2 $c2 | C5 $c5" | tee -a $O/sweep.txt
done
