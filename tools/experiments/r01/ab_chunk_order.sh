# GPU box: order in which (tile, frame) chunks are handed out: libdrt_o0 tile-major raster (default), o1 frame-major,
# o2 tile-major scattered over the image, o3 frame-major scattered
for wl in cornell_box_1080p_8spp_d8 suzanne_plane_1080p_8spp_d2 dense_monkey_1080p_16spp_d2 cs16_dust_1080p_8spp_d5 mc_transparency_843x460_50spp_d5; do for v in 0 1 4 5; do
  r=$(DRT_LIB_OVERRIDE=$PWD/dustraytracer_amd/libdrt_o$v.so python bench.py --workload $wl --cpu-seconds 0 --steps 90 --warmup 9 --no-roofline-counters 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$wl order=$v : $r ms/step"
done; done
for v in 0 1 4 5; do
  r=$(DRT_LIB_OVERRIDE=$PWD/dustraytracer_amd/libdrt_o$v.so python bench.py --emulate-shard 0/8 --cpu-seconds 0 --steps 240 --warmup 24 --no-roofline-counters 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "cornell 1/8 shard order=$v : $r ms/step"
done
