// render_kernels.hpp -- launch wrappers of render_kernels.hip (the seam that replaces InvokeRenderKernel,
// Core/Kernel/RenderKernel.cuh:12-14; grid/block shape is an internal choice here).
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <vector>

#include "device_scene.hpp"

namespace drt {

// pixel_walk: only in builds with -DDRT_WITH_PIXEL_WALK (`make pixel-walk`: the tests' cross-check library)
bool pixel_walk_built_in();
hipError_t launch_render(const SceneView &scene, const FrameParams &frame, int bvh_depth, bool count_work,
                         hipStream_t stream, const char **kernel_name);

// wave_queue (kernel_wave_queue.hip): persistent waves + tile queue + phase voting.
// mode 0 = lean (auto-upgraded to 1 when a setting or the scene needs it), 1 = general, 2 = general + work counters.
// which packaging of a wave_queue launch is fastest is measured, once per (kernel, scene shape, view class); one cache per renderer
struct WqVariant { int threads, entry_bytes, tris, per_cu; };
struct WqPlan { uint64_t key = 0; std::vector<WqVariant> cands; std::vector<double> ns_per_sample; std::vector<int> trials; int chosen = -1; };
struct WaveQueueCache { std::vector<WqPlan> plans; uint64_t batch_key = 0; int batch_cand = -1; double batch_samples = 0; };
void wave_queue_report(WaveQueueCache &cache, float span_ms);
WqVariant measured_choice(WaveQueueCache &cache, uint64_t key, double samples, const std::function<std::vector<WqVariant>()> &candidates);
constexpr size_t kLdsSceneBytes = 40 * 1024;     // stage the traversal data in LDS when it is at most this big
// `samples` must hold wave_queue_sample_bytes(frame) bytes (one float4 per pixel and frame of the launch); the launch
// runs the tracing kernel and then the ordered resolve kernel on `stream`.
size_t wave_queue_sample_bytes(const FrameParams &frame);
hipError_t launch_wave_queue(const SceneView &scene, const FrameParams &frame, int bvh_depth, int mode, bool scene_has_alpha,
                             unsigned int *chunk_counter, void *samples, int num_cus, hipStream_t stream, const char **kernel_name,
                             int *launch_shape /* out[4], may be null: stack slots per lane, workgroups per CU, LDS KiB per workgroup, threads per workgroup */,
                             WaveQueueCache &cache);

hipError_t launch_resolve(const FrameParams &frame, void *samples, hipStream_t stream);
size_t wave_queue_scene_lds_bytes(const SceneView &scene);

// The work-queue heads a launch draws from: one block of kPoolSampleShards counters, kPoolSampleShardStride words (128 bytes)
// apart (path_pool: sample ids sharded over them, so that a whole chip's waves do not serialise on one address; wave_queue
// uses the first word only).  The renderer hands out zeroed blocks.
#ifndef DRT_SAMPLE_SHARDS
#define DRT_SAMPLE_SHARDS 16          // (a power of two; -DDRT_SAMPLE_SHARDS=1 in EXTRA: the single counter of rounds 1-2, for A/B runs)
#endif
constexpr int kPoolSampleShards = DRT_SAMPLE_SHARDS, kPoolSampleShardStride = 32, kQueueHeadBlockWords = kPoolSampleShards * kPoolSampleShardStride;

// path_pool (kernel_path_pool.hip): path state parked in LDS, phase-homogeneous batches of 64 paths; lean paths of scenes
// whose traversal data fits LDS.  `status` is a device word the kernel sets when it had to abort (never hangs).
bool path_pool_supports(const SceneView &scene, const FrameParams &frame, int bvh_depth, size_t scene_lds_bytes, bool *hbm_scene);
void path_pool_leaf_classes(const std::vector<LeafRange> &leaves, uint32_t out[3]);
struct PoolTuning { int threads = 0, paths = 0, stack_lds = 0, min_fill = 48, patience = 8, n_loop = 8, n_min_lanes = 16, n_fuse_loop = 8, n_fuse_min = 24, cold_lds_kb = -1, share_grid = 1, dir_tries = 4; unsigned long long *stats = nullptr; };      // 0 = the launcher's default; stats: device u64[40] (DRT_POOL_STATS=1)
struct PoolScratch { void *aux = nullptr, *aux_slot = nullptr, *aux_light = nullptr, *aux_next = nullptr, *aux_stack = nullptr; size_t slots = 0, next_slots = 0, stack_slots = 0; };     // HBM part of the path state, owned by the renderer
hipError_t launch_path_pool(const SceneView &scene, const FrameParams &frame, int bvh_depth, bool scene_has_alpha, bool hbm_scene, const uint32_t t_class[3], const PoolTuning &tune,
                            PoolScratch &scratch, unsigned int *sample_counter, void *samples, unsigned int *status, int num_cus, hipStream_t stream, const char **kernel_name,
                            int *launch_shape /* out[5]: stack slots, workgroups per CU, LDS KiB, threads, pool paths */);

// debug: d_out2[0] += #floats in [first_bits, first_bits+count) where exact_rcp != 1.0f/x (which 0) or exact_sqrt != sqrtf (which 1),
// d_out2[1] += #floats on the fast path
hipError_t launch_check_rcp(int which, uint32_t first_bits, unsigned long long count, unsigned long long *d_out2, hipStream_t stream);

// debug: all values on pcg_hash cycles of length <= max_len; d_out = [count, (value, length) x cap_pairs]
hipError_t launch_hash_cycles(uint32_t max_len, uint32_t *d_out, uint32_t cap_pairs, hipStream_t stream);

// debug: device leaf functions on arrays (which: 0 unit vec, 1 unit sphere, 2 slab, 3 triangle, 4 camera ray, 5 unit disk)
hipError_t launch_kat(int which, const void *d_in, void *d_out, uint32_t n, const FrameParams &frame, hipStream_t stream);

hipError_t launch_assemble(const void *gathered, void *image, uint32_t width, uint32_t height, uint32_t stripe_rows,
                           uint32_t world, uint32_t padded_rows, hipStream_t stream);

}  // namespace drt
