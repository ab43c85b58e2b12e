for sh in 0/8 0/4 0/2 0/1; do
  r=$(python bench.py --emulate-shard $sh --cpu-seconds 0 --steps 300 --warmup 30 --no-roofline-counters | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "shard=$sh frames_in_flight=3 : $r"
done
python bench.py --frames-in-flight 1 --cpu-seconds 0 --no-roofline-counters | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('whole frame, 1 in flight:', d['ms_per_step'], d['value'])"
python tools/fixed_cost.py | grep -E "64x64|spp 1  samples   2073600|fit"
