# GPU box: 6-byte stack entries where they keep more waves resident (default) against 8-byte entries everywhere (DRT_STACK_REF16=0)
for wl in cs16_dust_1080p_8spp_d5 dense_monkey_1080p_16spp_d2 suzanne_plane_1080p_8spp_d2 mc_transparency_843x460_50spp_d5; do for rep in 1 2; do for t in 0 1; do
  r=$(DRT_STACK_REF16=$t python bench.py --workload $wl --cpu-seconds 0 --steps 60 --warmup 6 --no-roofline-counters 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'] and d['roofline'].get('kernel') or '')")
  echo "$wl ref16=$t : $r"
done; done; done
for t in 0 1; do DRT_STACK_REF16=$t python tools/time_workload.py cs16_dust 1920 1080 8 2>/dev/null | tail -1; done
