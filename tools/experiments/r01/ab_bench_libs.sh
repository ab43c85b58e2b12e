# GPU box: A/B builds of the library with the bench itself, every workload:  bash tools/ab_bench_libs.sh libdrt_a.so libdrt_b.so ...
for wl in cornell_box_1080p_8spp_d8 suzanne_plane_1080p_8spp_d2 dense_monkey_1080p_16spp_d2 cs16_dust_1080p_8spp_d5 mc_transparency_843x460_50spp_d5; do for rep in 1 2; do for l in "$@"; do
  r=$(DRT_LIB_OVERRIDE=$PWD/dustraytracer_amd/$l python bench.py --workload $wl --cpu-seconds 0 --steps 90 --warmup 9 --no-roofline-counters 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$wl $l : $r ms/step"
done; done; done
for l in "$@"; do
  r=$(DRT_LIB_OVERRIDE=$PWD/dustraytracer_amd/$l python bench.py --emulate-shard 0/8 --cpu-seconds 0 --steps 240 --warmup 24 --no-roofline-counters 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "cornell 1/8 shard $l : $r ms/step"
done
