# GPU box: T-step leaf chaining forced off / on / by depth (default), per scene
for wl in "cornell_box 1920 1080 8" "suzanne_plane 1920 1080 8" "room 1920 1080 8" "cs16_dust 1920 1080 8" "mc_transparency 1920 1080 8"; do
  for c in 0 1 -1; do
    r=$(DRT_LEAF_CHAIN=$c timeout -k 10 100 python tools/time_workload.py $wl | tail -1 | sed 's/.*ms \([0-9.]*\) wall.*s  \(.*\)/\1 ms \2/')
    echo "$wl chain=$c : $r"
  done
done
