for sc in suzanne_plane dense_monkey mc_transparency cs16_dust room; do
  for v in "12 36 4" "16 36 4" "16 44 6" "20 44 6"; do set -- $v
    r=$(DRT_VOTE_N=$1 DRT_VOTE_S=$2 DRT_VOTE_R=$3 timeout -k 10 100 python tools/time_workload.py $sc 1920 1080 8 | tail -1 | sed 's/.*ms \([0-9.]*\) wall.*s  \(.*\)/\1 ms \2/')
    echo "$sc N=$1 S=$2 R=$3 : $r"
  done
done
