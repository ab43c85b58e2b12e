# GPU box: three triangles per T step where launch_one chooses them (default) against two everywhere (DRT_TRIS_WIDE=0),
# on the workloads whose scene is read from HBM
for wl in suzanne_plane_1080p_8spp_d2 dense_monkey_1080p_16spp_d2 cs16_dust_1080p_8spp_d5 mc_transparency_843x460_50spp_d5; do for rep in 1 2; do for w in 0 1; do
  r=$(DRT_TRIS_WIDE=$w python bench.py --workload $wl --cpu-seconds 0 --steps 60 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel'])")
  echo "$wl wide=$w : $r"
done; done; done
