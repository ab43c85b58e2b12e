# GPU box: 512-thread workgroups where they keep more waves resident (default) against 256 everywhere (DRT_WG_THREADS=256)
for wl in "room 1920 1080 8" "cornell_box 1920 1080 8" "sunshadow_test 1920 1080 8" "lightweight_rt 1920 1080 8"; do for t in 256 0; do
  r=$(DRT_WG_THREADS=$t timeout -k 10 100 python tools/time_workload.py $wl 2>/dev/null | tail -1 | sed 's/.*depth [0-9]* \(.*\) wall.*/\1/')
  echo "$wl wg=$t : $r"
done; done
for t in 256 0; do
  r=$(DRT_WG_THREADS=$t python bench.py --workload room_4k_64spp_d16 --cpu-seconds 0 --steps 3 --warmup 1 --no-roofline-counters 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "room_4k_64spp_d16 wg=$t : $r"
done
