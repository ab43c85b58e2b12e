# GPU box: everything profiles/ is built from.  Output under gpurun_out/prof/ (tools/install_profiles.py copies what is kept).
#   bash tools/collect_profiles.sh [round-tag] [counters|bench|all] [workload ...]
# counters: per workload, one `bench.py --frames-in-flight 1` run each under rocprofv3 --kernel-trace --stats and under six --pmc
#           passes (SQ instruction mix; SQ wave time; L2 requests / hits; vector-memory + LDS instructions; FETCH_SIZE; WRITE_SIZE --
#           separate passes, MI355X_MICROARCH.md "HBM"), then tools/roofline_from_profiles.py -> <tag>_roofline_<workload>.json
# bench:    the bench lines of every workload (they read the roofline JSONs written above), emulated shards, pool statistics
set -e
TAG=${1:-r03}; WHAT=${2:-all}; shift || true; shift || true
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
WLS="$@"
[ -z "$WLS" ] && WLS="cornell_box_1080p_8spp_d8 cornell_box_256_1spp_d4 suzanne_plane_1080p_8spp_d2 dense_monkey_1080p_16spp_d2 room_4k_64spp_d16 cs16_dust_1080p_8spp_d5"
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32"
SQ2="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
SQ3="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
L2="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum"
if [ "$WHAT" != "bench" ]; then
for wl in $WLS; do
  steps=6; warm=1; [ $wl = room_4k_64spp_d16 ] && steps=2
  [ $wl = cornell_box_256_1spp_d4 ] && steps=40
  B="python3 $R/bench.py --workload $wl --frames-in-flight 1 --cpu-seconds 0 --steps $steps --warmup $warm --no-roofline-counters"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kt_$wl -o kt -- $B > $O/${TAG}_bench_f1_under_rocprof_$wl.json 2> /tmp/p_kt.log
  cp /tmp/p_kt_$wl/kt_kernel_stats.csv $O/${TAG}_bench_f1_kernel_stats_$wl.csv
  rocprofv3 --pmc $SQ1 --output-format csv -d /tmp/p_sq1_$wl -o sq -- $B > /dev/null 2> /tmp/p_sq1.log
  rocprofv3 --pmc $SQ2 --output-format csv -d /tmp/p_sq2_$wl -o sq -- $B > /dev/null 2> /tmp/p_sq2.log
  rocprofv3 --pmc $SQ3 --output-format csv -d /tmp/p_sq3_$wl -o sq -- $B > /dev/null 2> /tmp/p_sq3.log
  rocprofv3 --pmc $L2 --output-format csv -d /tmp/p_l2_$wl -o l2 -- $B > /dev/null 2> /tmp/p_l2.log || echo "L2 pass failed for $wl (see /tmp/p_l2.log)"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch_$wl -o fetch -- $B > /dev/null 2> /tmp/p_fetch.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_write_$wl -o write -- $B > /dev/null 2> /tmp/p_write.log
  for d in sq1 sq2 sq3 l2 fetch write; do
    [ -d /tmp/p_${d}_$wl ] || continue
    python3 $R/tools/pmc_summary.py /tmp/p_${d}_$wl > $O/${TAG}_pmc_${d}_$wl.txt || true
    f=$(find /tmp/p_${d}_$wl -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $R/tools/trim_counter_csv.py $f $O/${TAG}_pmc_${d}_$wl.csv
  done
  python3 $R/tools/roofline_from_profiles.py $wl $O/${TAG}_bench_f1_kernel_stats_$wl.csv /tmp/p_sq1_$wl /tmp/p_sq2_$wl /tmp/p_sq3_$wl /tmp/p_l2_$wl /tmp/p_fetch_$wl /tmp/p_write_$wl --out $O/${TAG}_roofline_$wl.json > /dev/null
  echo "counters done: $wl"
done
fi
if [ "$WHAT" != "counters" ]; then
# the bench lines (they read the roofline JSONs written above: install them first for this run)
cp $O/${TAG}_roofline_*.json $R/profiles/ 2>/dev/null || true
cd $R
python3 bench.py > $O/${TAG}_bench_default.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_f3 -o f3 -- python3 $R/bench.py --cpu-seconds 0 > $O/${TAG}_bench_default_under_rocprof.json 2> /tmp/p_f3.log; cp /tmp/p_f3/f3_kernel_stats.csv $O/${TAG}_bench_default_kernel_stats.csv)
python3 bench.py --workload room_4k_64spp_d16 --cpu-seconds 10 --steps 4 --warmup 1 > $O/${TAG}_bench_room_4k_64spp_d16.json
for wl in suzanne_plane_1080p_8spp_d2 dense_monkey_1080p_16spp_d2 cs16_dust_1080p_8spp_d5 cornell_box_256_1spp_d4 mc_transparency_843x460_50spp_d5; do
  python3 bench.py --workload $wl --cpu-seconds 3 > $O/${TAG}_bench_$wl.json
done
for s in 0/2 0/4 0/8; do python3 bench.py --emulate-shard $s --cpu-seconds 0 --steps 200 --warmup 20 > $O/${TAG}_bench_shard_${s/\//of}.json; done
python3 tools/pool_stats.py cornell_box 1920 1080 8 8 > $O/${TAG}_pool_stats_cornell.txt 2>&1
python3 tools/pool_stats.py room 1920 1080 4 16 > $O/${TAG}_pool_stats_room.txt 2>&1
python3 tools/small_launches.py > $O/${TAG}_small_launches.txt 2>&1
fi
ls $O | wc -l
