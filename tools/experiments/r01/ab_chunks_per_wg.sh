# GPU box: chunks per workgroup (grid size of small launches) against step time, shards and whole frame, 3 frames in flight
for k in 4 32 64 96 128; do for sh in 0/8 0/4 0/2 0/1; do
  r=$(DRT_CHUNKS_PER_WG=$k python bench.py --emulate-shard $sh --cpu-seconds 0 --steps 300 --warmup 30 --no-roofline-counters | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "chunks/wg=$k shard=$sh : $r ms/step"
done; done
for k in 4 64; do echo "chunks/wg=$k isolated 1-spp 1080p: $(DRT_CHUNKS_PER_WG=$k python tools/fixed_cost.py | grep -E 'spp 1  samples   2073600|fit')"; done
