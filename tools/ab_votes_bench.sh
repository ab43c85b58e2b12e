# GPU box: shade-vote threshold measured with the bench itself (3 frames in flight), then on the other scenes
for rep in 1 2; do for s in 36 40 42 44 46 48; do
  r=$(DRT_VOTE_S=$s python bench.py --cpu-seconds 0 --steps 180 --warmup 12 --no-roofline-counters 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "rep=$rep cornell bench S=$s : $r ms/step"
done; done
for wl in suzanne_plane_1080p_8spp_d2 dense_monkey_1080p_16spp_d2 cs16_dust_1080p_8spp_d5 mc_transparency_843x460_50spp_d5 cornell_box_256_1spp_d4; do for s in 36 40 44 48; do
  r=$(DRT_VOTE_S=$s python bench.py --workload $wl --cpu-seconds 0 --steps 60 --warmup 6 --no-roofline-counters 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$wl S=$s : $r ms/step"
done; done
for s in 36 44; do r=$(DRT_VOTE_S=$s python tools/time_workload.py room 1920 1080 8 2>/dev/null | tail -1 | sed 's/.*ms \([0-9.]*\) wall.*/\1/'); echo "room 1080p S=$s : $r ms"; done
for s in 36 44; do r=$(DRT_VOTE_S=$s python bench.py --emulate-shard 0/8 --cpu-seconds 0 --steps 240 --warmup 24 --no-roofline-counters 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"); echo "shard 0/8 S=$s : $r ms/step"; done
