# GPU box: frames in flight against step time (whole frame and 1/8 shard)
for sh in 0/1 0/8; do for f in 1 2 3 4 5 6 8; do
  r=$(python bench.py --emulate-shard $sh --frames-in-flight $f --cpu-seconds 0 --steps 240 --warmup 24 --no-roofline-counters | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "shard=$sh frames_in_flight=$f : $r ms/step"
done; done
