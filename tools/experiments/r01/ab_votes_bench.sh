# GPU box: one voting knob at a time, measured with the bench itself (whole frame, 3 in flight; and the 1/8 shard)
#   bash tools/ab_votes_bench.sh
run() { python bench.py "$@" --cpu-seconds 0 --no-roofline-counters 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for kv in "DRT_VOTE_S=44" "DRT_VOTE_R=2" "DRT_VOTE_R=6" "DRT_VOTE_R=8" "DRT_VOTE_P=4" "DRT_VOTE_P=12" "DRT_VOTE_P=16" "DRT_VOTE_TN=2" "DRT_VOTE_TN=8" "DRT_VOTE_TS=20" "DRT_VOTE_TS=44" "DRT_VOTE_N=10" "DRT_VOTE_N=14" "DRT_VOTE_S=44"; do
  a=$(env $kv python bench.py --cpu-seconds 0 --no-roofline-counters --steps 180 --warmup 12 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(env $kv python bench.py --cpu-seconds 0 --no-roofline-counters --emulate-shard 0/8 --steps 240 --warmup 24 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$kv : frame $a ms/step, 1/8 shard $b ms/step"
done
