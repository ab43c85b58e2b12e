# GPU box: everything profiles/ is built from, in one call.  Output under gpurun_out/prof/ (copy what is to be kept).
#   bash tools/collect_profiles.sh
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the default bench line, and its kernel trace (same command under rocprofv3)
python3 $R/bench.py > $O/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_f3 -o f3 -- python3 $R/bench.py --cpu-seconds 0 > $O/bench_default_under_rocprof.json 2> /tmp/p_f3.log
cp /tmp/p_f3/f3_kernel_stats.csv $O/bench_default_f3_kernel_stats.csv
# 2. isolated launches
python3 $R/bench.py --frames-in-flight 1 --cpu-seconds 0 > $O/bench_f1_isolated.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_f1 -o f1 -- python3 $R/bench.py --frames-in-flight 1 --cpu-seconds 0 > $O/bench_f1_under_rocprof.json 2> /tmp/p_f1.log
cp /tmp/p_f1/f1_kernel_stats.csv $O/bench_f1_isolated_kernel_stats.csv
# 3. HBM traffic, separate passes per counter (isolated launches so that a dispatch is one launch)
for wl in cornell_box_1080p_8spp_d8 cs16_dust_1080p_8spp_d5; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch_$wl -o fetch -- python3 $R/bench.py --workload $wl --frames-in-flight 1 --cpu-seconds 0 --steps 5 --warmup 1 --no-roofline-counters > /dev/null 2> /tmp/p_fetch.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_write_$wl -o write -- python3 $R/bench.py --workload $wl --frames-in-flight 1 --cpu-seconds 0 --steps 5 --warmup 1 --no-roofline-counters > /dev/null 2> /tmp/p_write.log
  python3 $R/tools/pmc_traffic.py /tmp/p_fetch_$wl /tmp/p_write_$wl $wl $O/traffic_$wl.json > /dev/null
  cp /tmp/p_fetch_$wl/fetch_counter_collection.csv $O/pmc_fetch_size_$wl.csv
  cp /tmp/p_write_$wl/write_counter_collection.csv $O/pmc_write_size_$wl.csv
done
# 4. SQ counters of the tracing kernel
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/p_sq1 -o sq -- python3 $R/tools/time_workload.py cornell_box 1920 1080 8 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS --output-format csv -d /tmp/p_sq2 -o sq -- python3 $R/tools/time_workload.py cornell_box 1920 1080 8 > /dev/null 2>&1
python3 $R/tools/pmc_summary.py /tmp/p_sq1 /tmp/p_sq2 > $O/wave_queue_pmc_sq.txt
# 5. phase statistics (counting build) and the other workloads
python3 $R/tools/phase_stats.py cornell_box 8 > $O/phase_stats_cornell.txt
python3 $R/tools/phase_stats.py cs16_dust 8 > $O/phase_stats_cs16_dust.txt
for wl in suzanne_plane_1080p_8spp_d2 dense_monkey_1080p_16spp_d2 cs16_dust_1080p_8spp_d5 cornell_box_256_1spp_d4 mc_transparency_843x460_50spp_d5; do
  python3 $R/bench.py --workload $wl --cpu-seconds 3 > $O/bench_$wl.json
done
python3 $R/bench.py --workload room_4k_64spp_d16 --cpu-seconds 3 --steps 3 --warmup 1 > $O/bench_room_4k_64spp_d16.json
for s in 0/2 0/4 0/8; do python3 $R/bench.py --emulate-shard $s --cpu-seconds 0 --steps 200 --warmup 20 > $O/bench_shard_${s/\//of}.json; done
# 6. BVH build on the device: host vs GPU table, and the per-kernel times of the same command
cd $R && python3 tools/bvh_build_bench.py > $O/bvh_build_host_vs_gpu.txt 2> /dev/null
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bvh -o bvh -- python3 $R/tools/bvh_build_bench.py > /dev/null 2> /tmp/p_bvh.log
cp /tmp/p_bvh/bvh_kernel_stats.csv $O/bvh_build_kernel_stats.csv
ls -la $O
