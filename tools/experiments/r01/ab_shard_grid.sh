# GPU box: 1/8 shard step time against workgroups per CU and frames in flight
for b in 6 4 3 2; do for f in 2 3 4 6; do
  r=$(DRT_MAX_BLOCKS_PER_CU=$b python bench.py --emulate-shard 0/8 --frames-in-flight $f --cpu-seconds 0 --steps 300 --warmup 30 --no-roofline-counters | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "wg/CU<=$b frames_in_flight=$f : $r ms/step"
done; done
