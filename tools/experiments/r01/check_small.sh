for wl in cornell_box_256_1spp_d4 cornell_box_1080p_8spp_d8; do
  python bench.py --workload $wl --cpu-seconds 0 --steps 300 --warmup 30 --no-roofline-counters | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['ms_per_step'], d['value'])"
done
for sh in 0/8 0/4 0/2; do
  python bench.py --emulate-shard $sh --cpu-seconds 0 --steps 300 --warmup 30 --no-roofline-counters | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('shard $sh', d['ms_per_step'], d['value'])"
done
