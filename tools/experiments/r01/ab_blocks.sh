# GPU box: sensitivity to resident workgroups per CU
for wl in "cs16_dust 1920 1080 8" "suzanne_plane 1920 1080 8" "cornell_box 1920 1080 8"; do
  for b in 2 3 4 5 6 8; do
    r=$(DRT_MAX_BLOCKS_PER_CU=$b timeout -k 10 100 python tools/time_workload.py $wl | tail -1 | sed 's/.*ms \([0-9.]*\) wall.*/\1/')
    echo "$wl blocks/CU<=$b : $r ms"
  done
done
