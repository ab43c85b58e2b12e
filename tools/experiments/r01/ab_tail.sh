# GPU box: drain-phase vote thresholds (DRT_VOTE_TS / DRT_VOTE_TN) against the fixed cost of a launch
for cfg in "36 12" "16 8" "8 4" "4 2" "1 1" "8 12" "36 4"; do set -- $cfg
  echo "== tail S>=$1 N>=$2"
  DRT_VOTE_TS=$1 DRT_VOTE_TN=$2 python tools/fixed_cost.py | grep -E "64x64|1920x136|spp 1 .*2073600|spp 8  samples  16588800|fit"
done
