// kernel_path_pool.hip -- "path_pool": the tracing kernel.  The PATH STATE is parked in LDS and every phase is run for 64 paths
// that all need exactly that phase; the traversal data lives in LDS too where it fits (lds-scene build), else it is read from
// global memory (hbm-scene build).
//
// wave_queue (kernel_wave_queue.hip) keeps one path per lane in registers and lets the 64 lanes of a wave vote on the
// phase to run next; its counters say half of every vector instruction is masked-off lanes (profiles/
// r01_wave_queue_pmc_sq.txt: lane utilisation 0.51) because at any moment the lanes of a wave want different things.
// Here a lane owns nothing.  A workgroup keeps a POOL of P paths (P ~ 1-1.4 x its lanes, as many as LDS holds), and every
// path waits in exactly one queue:
//     N      pop entries of its traversal stack; interior node: slab-test both children, push       BVHTraversal.cuh:33-72
//     T0..T3 test the triangles of the leaf it stands on, two (hbm-scene: four) per step -- one queue per
//            leaf-size class, so that the lanes of a batch run the same number of steps              BVHTraversal.cuh:46-57
//     B      shade the closest hit, draw the bounce direction (a few candidates), launch the bounce ray  RayGen.cuh:90-134
//            (sunlight builds: launch the sun's shadow ray instead -- a traversal through N and T without culls whose
//            first accepted hit ends it, BVHTraversal.cuh:76-134)
//     S      sunlight builds: after the shadow traversal, add the sunlight, then the direction and the bounce ray  RayGen.cuh:126-134
//     R      more candidates for the paths whose direction was still rejected, then launch             Random.cu:50-58
//     E      finish the path (sky term, tone map, gamma, store the sample), take a new sample,
//            generate its primary ray                                                               RayGen.cuh:63-108,165-171
// A wave claims up to 64 path ids of ONE queue, loads the part of the state that phase needs, runs the phase with every
// lane busy, stores what changed and pushes each id to the queue of its next phase.  State per path: 36 bytes of LDS
// (A {origin, hit distance}, B {direction, leaf range | stack height}, W {hit triangle | bounce | flags}; 48 bytes with
// 32-bit triangle indices in the hbm-scene build) + its traversal stack (8 bytes per BVH level below the root; 6 in the
// hbm-scene build); throughput, RNG state, the sample's slot and the light gathered so far -- touched by the shading phases
// only -- live in HBM.  Per-lane order of node visits, triangle tests and RNG draws is the reference's, and the arithmetic is the
// same device_math.hpp code as wave_queue's, so the image is bit-identical; only who computes what when differs.
//
// Queues are rings of 16-bit entries in LDS, multi-producer / multi-consumer inside the workgroup.  An entry is
// path id (12 bits, P <= 4032) | lap of its ring position (position / ring capacity, 4 bits) << 12; 0xFFFF = empty.
// A producer reserves a position with an atomic add on the tail and writes its entry there once the slot is empty; a consumer
// claims [head, head+n) with one compare-and-swap on the head, reads the entries -- accepting only an entry that carries the
// lap of ITS position (anything else is re-read: empty = reserved but not written yet, another lap's tag = the entry of the
// position one lap earlier, whose consumer has claimed it but not taken it yet) -- and empties the slots.  Both directions are
// closed that way: a producer never overwrites an entry that has not been taken (round 2 had that), and a consumer never takes
// the entry of an earlier lap (round 2 did not: a wave stalled between its claim and its read while the other paths cycled
// the ring once could see its entry taken twice -- one path in two queues; tools/sim_queue.py --ring-hazard replays it).
// Every wait is on a strictly older ring position, so there is no cycle.
// Every wait in this kernel is bounded: a wave that polls too long raises the abort flag, all waves leave, the host
// reports DRT_ERR_DEVICE (status word) -- a logic error must never hang the GPU; the hbm-scene build also clamps every
// index it addresses global memory with (a violation is reported the same way instead of faulting).
// No MFMA (branchy scalar fp32 / u32).  Citations are relative to /root/reference/DustRayTracer/src/.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "device_access.hpp"
#include "device_math.hpp"
#include "device_scene.hpp"
#include "render_kernels.hpp"

namespace drt {

namespace {

constexpr int kNQ = 9;                               // queues: N, T0..T3, B, E, R, S
enum : int { QN = 0, QT0 = 1, QB = 5, QE = 6, QR = 7, QS = 8 };
constexpr uint32_t kEmptyId = 0xFFFFu;                // ring entry: path id | (lap & 15) << 12; ids stop at 4031, so 0xFFFF is never an entry
constexpr uint32_t kIdMask = 0xFFFu;
[[maybe_unused]] constexpr uint32_t kMaxPoolPaths = 4032u;
constexpr uint32_t kSampleShards = (uint32_t)kPoolSampleShards, kSampleShardStride = (uint32_t)kPoolSampleShardStride;     // (render_kernels.hpp)
constexpr uint32_t kNoPrim = 0xFFFu;                 // word W: hit triangle (12 bits, kNoPrim = none) | bounce index << 12 (16 bits) | kHasSample
constexpr uint32_t kHasSample = 1u << 28;
constexpr uint32_t kShadow = 1u << 29;               // the traversal under way is the sun's shadow ray (RayTest, BVHTraversal.cuh:76-134)
constexpr uint32_t kBackFace = 1u << 30;             // the shaded hit was seen from behind: its normal is -face normal (ClosestHit.cuh:17-24)
constexpr uint32_t kOccluded = 1u << 31;             // the shadow ray hit something
constexpr uint32_t kMetal = 1u << 27;                // material-model builds: the direction being drawn is the mirror lobe's (R finds roughness and normal again from the hit triangle)
constexpr int kMaxPoolThreads = 1024;                // up to 16 waves per workgroup = 4 per SIMD (128 VGPRs each)
// The hbm-scene production kernels are built NARROW: at most 12 waves per workgroup and 80 VGPRs (6 waves per SIMD), two triangles per
// T step instead of four -- so that TWO workgroups are resident per CU, 24 waves instead of 16.  Those builds wait for L2 ~43 % of
// their cycles; half as many loads in flight per lane and half as many more waves is the better trade: dense_monkey 10 519 -> 11 800
// Msamples/s, suzanne 11 750 -> 13 000, cs16_dust 1 230 -> 1 310 (tools/experiments/r03/patches/hbm_two_pools.patch is the experiment).
// Held to 80 VGPRs WITH four triangles per step the same kernel spills 25 of them and loses 7-14 %.  (A workgroup's waves must divide by
// four: two workgroups of 10 waves -- 96 VGPRs -- do not fit side by side, 3 + 3 + 2 + 2 on the SIMDs twice.)  The statistics builds
// carry eleven more counters per lane and stay wide: one workgroup of 16 waves.
constexpr bool pool_narrow(int flags) { return (flags & 8) != 0 && (flags & 1) == 0; }
constexpr int pool_max_threads(int flags) { return pool_narrow(flags) ? 768 : kMaxPoolThreads; }
constexpr int pool_min_waves(int flags) { return pool_narrow(flags) ? 6 : 4; }

typedef uint32_t pp_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t pp_u32x2 __attribute__((ext_vector_type(2)));
#define PP_LDS(T) __attribute__((address_space(3))) T
#if defined(__HIP_DEVICE_COMPILE__)
DRT_DEV uint4 ld4(uint32_t off) { const pp_u32x4 v = *(PP_LDS(const pp_u32x4) *)off; return make_uint4(v.x, v.y, v.z, v.w); }
DRT_DEV uint2 ld2(uint32_t off) { const pp_u32x2 v = *(PP_LDS(const pp_u32x2) *)off; return make_uint2(v.x, v.y); }
DRT_DEV uint32_t ld1(uint32_t off) { return *(PP_LDS(const uint32_t) *)off; }
DRT_DEV void st4(uint32_t off, uint4 v) { pp_u32x4 w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; *(PP_LDS(pp_u32x4) *)off = w; }
DRT_DEV void st2(uint32_t off, uint2 v) { pp_u32x2 w; w.x = v.x; w.y = v.y; *(PP_LDS(pp_u32x2) *)off = w; }
DRT_DEV void st1(uint32_t off, uint32_t v) { *(PP_LDS(uint32_t) *)off = v; }
DRT_DEV uint32_t ld_u16(uint32_t off) { return *(PP_LDS(const unsigned short) *)off; }
DRT_DEV void st_u16(uint32_t off, uint32_t v) { *(PP_LDS(unsigned short) *)off = (unsigned short)v; }
// control words and ring entries are shared between waves: relaxed atomics (never cached in registers), workgroup scope
DRT_DEV uint32_t ld1_shared(uint32_t off) { return __hip_atomic_load((PP_LDS(uint32_t) *)off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DRT_DEV void st1_shared(uint32_t off, uint32_t v) { __hip_atomic_store((PP_LDS(uint32_t) *)off, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DRT_DEV uint2 ld2_shared(uint32_t off) {
    const unsigned long long v = __hip_atomic_load((PP_LDS(unsigned long long) *)off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
DRT_DEV uint32_t ld_id(uint32_t off) { return __hip_atomic_load((PP_LDS(unsigned short) *)off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DRT_DEV void st_id(uint32_t off, uint32_t v) { __hip_atomic_store((PP_LDS(unsigned short) *)off, (unsigned short)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DRT_DEV uint32_t lds_add(uint32_t off, uint32_t v) { return __hip_atomic_fetch_add((PP_LDS(uint32_t) *)off, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DRT_DEV bool lds_cas(uint32_t off, uint32_t expect, uint32_t desired) {
    return __hip_atomic_compare_exchange_strong((PP_LDS(uint32_t) *)off, &expect, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// compare-and-swap that also reports the value found (the new head, when another wave won the race)
DRT_DEV bool lds_cas_seen(uint32_t off, uint32_t expect, uint32_t desired, uint32_t &seen) {
    const bool ok = __hip_atomic_compare_exchange_strong((PP_LDS(uint32_t) *)off, &expect, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    seen = expect;
    return ok;
}
DRT_DEV void lds_add64(uint32_t off, unsigned long long v) { (void)__hip_atomic_fetch_add((PP_LDS(unsigned long long) *)off, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DRT_DEV unsigned long long ld64(uint32_t off) { return *(PP_LDS(const unsigned long long) *)off; }
DRT_DEV void lds_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
DRT_DEV void lds_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
#else       // host pass: never called, only parsed
__device__ inline uint4 ld4(uint32_t) { return make_uint4(0, 0, 0, 0); }
__device__ inline uint2 ld2(uint32_t) { return make_uint2(0, 0); }
__device__ inline uint32_t ld1(uint32_t) { return 0; }
__device__ inline void st4(uint32_t, uint4) {}
__device__ inline void st2(uint32_t, uint2) {}
__device__ inline void st1(uint32_t, uint32_t) {}
__device__ inline uint32_t ld_u16(uint32_t) { return 0; }
__device__ inline void st_u16(uint32_t, uint32_t) {}
__device__ inline uint32_t ld1_shared(uint32_t) { return 0; }
__device__ inline void st1_shared(uint32_t, uint32_t) {}
__device__ inline uint2 ld2_shared(uint32_t) { return make_uint2(0, 0); }
__device__ inline uint32_t ld_id(uint32_t) { return 0; }
__device__ inline void st_id(uint32_t, uint32_t) {}
__device__ inline uint32_t lds_add(uint32_t, uint32_t) { return 0; }
__device__ inline bool lds_cas(uint32_t, uint32_t, uint32_t) { return false; }
__device__ inline bool lds_cas_seen(uint32_t, uint32_t, uint32_t, uint32_t &) { return false; }
__device__ inline void lds_add64(uint32_t, unsigned long long) {}
__device__ inline unsigned long long ld64(uint32_t) { return 0; }
__device__ inline void lds_release() {}
__device__ inline void lds_acquire() {}
#endif
DRT_DEV unsigned long long pp_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
DRT_DEV int pp_rank(unsigned long long mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
DRT_DEV float u2f(uint32_t u) { return __uint_as_float(u); }
DRT_DEV uint32_t f2u(float f) { return __float_as_uint(f); }

// LDS map of a workgroup (byte offsets from the start of its dynamic LDS; host and device compute it the same way)
struct PoolLayout {
    uint32_t ctrl, rings, quads, words, stack, scene, cold, total;      // byte offsets; total = bytes needed
};
__host__ __device__ inline PoolLayout pool_layout(uint32_t P, uint32_t ring_cap, uint32_t stack_entries, uint32_t scene_bytes, uint32_t cold_bytes,
                                                  uint32_t n_rings = (uint32_t)kNQ - 1u /* S's ring exists in sunlight builds only */,
                                                  uint32_t word_bytes = 4u /* 16 in the hbm-scene build: {meta, hit triangle, leaf end, -} */,
                                                  uint32_t stack_entry_bytes = 8u /* 6 in the hbm-scene build */,
                                                  uint32_t stats_bytes = 0u /* statistics build: per-queue {batches, lanes, ticks} sums of the workgroup */) {
    PoolLayout l;
    l.ctrl = 0;                                     // head/tail pairs of the kNQ queues (8 B each), then {live, abort}, {exhausted, -}
    l.rings = 128 + stats_bytes;                    // (the statistics block sits between the control words and the rings)
    l.quads = l.rings + n_rings * ring_cap * 2u;
    l.words = l.quads + 2u * P * 16u;
    l.stack = l.words + P * word_bytes;
    l.scene = (l.stack + stack_entries * P * stack_entry_bytes + 15u) & ~15u;
    l.cold = l.scene + scene_bytes + (scene_bytes ? 48u : 0u);      // (T reads pairs of triangles: the partner of the last one is one record past the end)
    l.total = l.cold + cold_bytes;
    return l;
}
constexpr uint32_t kCtrlLive = 80, kCtrlAbort = 84, kCtrlExhausted = 88, kCtrlStats = 128, kStatsBytes = 256;      // (the pairs of lanes 10 and 11 of the control read)
__host__ __device__ inline uint32_t pool_scene_bytes(const SceneView &sc) { return sc.n_inner * 64u + sc.n_tris * 48u; }   // (leaf ranges ride in the references)
__host__ __device__ inline uint32_t pool_cold_bytes(const SceneView &sc) { return sc.n_tris * 32u + sc.n_mats * 16u + sc.n_texs * 16u; }

// What the host decides per launch (next to FrameParams)
struct PoolParams {
    uint32_t P;                // paths in the pool (multiple of 64)
    uint32_t ring_cap;         // entries per ring: power of two >= P
    uint32_t ring_shift;       // log2(ring_cap): ring position >> ring_shift = the lap, whose low 4 bits tag the entry
    uint32_t stack_entries;    // BVH levels - 1
    uint32_t stack_lds;        // of which this many (the stack's bottom levels) are kept in LDS; the levels above them in HBM (aux_stack)
    uint32_t total_samples;    // n_chunks * 64 (sample ids beyond the image edge are skipped)
    uint32_t n_chunks, tiles_x;
    uint32_t t_class[3];       // leaf step counts (two triangles per step) up to t_class[i] wait in queue T<i>; larger ones in T3
    uint32_t min_fill;         // a wave prefers waiting to running a batch thinner than this ...
    uint32_t patience;         // ... for this many polls
    uint32_t n_loop;           // N: at most this many pops per batch ...
    uint32_t n_min_lanes;      // ... and the batch ends when fewer lanes than this are still popping (the rest go back to N)
    uint32_t n_fuse_loop;      // a batch that launches rays (B / S / R / E): at most this many visits per new ray, the root's included ...
    uint32_t n_fuse_min;       // ... and only while at least this many lanes still need one
    uint32_t cold_in_lds;      // TriCold / MatDev / TexDev records staged in LDS too (small scenes): B's loads chain through LDS
    uint32_t dir_tries;        // B and R draw at most this many candidates of the bounce direction per batch; paths still without one go (back) to R
    uint4 *aux;                // HBM, [workgroup][path]: {throughput, seed} -- what only B and E touch stays out of LDS
    float4 *aux_light;         // HBM, [workgroup][path]: light gathered so far (sunlight builds; lean paths gather light only where they end)
    uint32_t *aux_slot;        // HBM, [workgroup][path]: where the path's sample goes in `samples`
    uint2 *aux_stack;          // HBM, [workgroup][level - stack_lds][path]: the stack entries above the levels kept in LDS (deep trees)
    float4 *aux_next;          // HBM, [workgroup][path][2]: material-model builds with sunlight: what B decided about the continuation ray, kept across the shadow traversal
    unsigned int *status;      // device word: != 0 after an aborted launch
    unsigned long long *stats; // STATS build: per queue {batches, lanes, ticks} (3 x kNQ), then claim ticks, idle polls, lost claims, wave ticks
};

// FLAGS: 1 = statistics, 2 = sunlight (a shadow ray per shaded hit, RayGen.cuh:124-128), 4 = alpha cut-outs (AnyHit.cuh:8-28),
// 8 = hbm-scene: the traversal data is too big for LDS and is read from global memory (L2); the pool -- path state, stacks, queues --
// is the same, with 32-bit triangle indices: word W becomes a quad {meta = bounce | stack height << 16 | flags, hit triangle, leaf end, -}
// and B's fourth word the current triangle; 16 = the opt-in material model (kernel_path_pool_ext.o);
// 32 = deep tree: only the bottom PoolParams::stack_lds levels of the traversal stacks are in LDS, the levels above them in HBM
// (launch_path_pool chooses: the LDS that frees holds more paths, or a second workgroup per CU)
// The arguments travel as ONE struct, and the kernel never names it: every field is read from the kernarg segment where it is
// used, through a pointer that is made opaque at the top of every batch (`ka`, below).  Named by-value arguments are loaded at
// the kernel's entry and stay live across the whole loop -- ~130 dwords of them, which the register allocator then spills to
// VGPR lanes and reloads inside the claim, push and traversal code (85 SGPR spills in round 2); read where they are used, the
// camera, sky and frame constants occupy registers only inside the phase that needs them.
struct PoolArgs { SceneView sc; FrameParams fp; PoolParams pp; unsigned int *sample_counter; float4 *samples; };
template <int FLAGS>
__global__ __launch_bounds__(pool_max_threads(FLAGS)) __attribute__((amdgpu_waves_per_eu(pool_min_waves(FLAGS)))) void path_pool_kernel(const PoolArgs) {
    // the kernel arguments, read from the kernarg segment (scalar loads) at the point of use
    typedef const PoolArgs __attribute__((address_space(4))) *KernArgs;
    KernArgs ka = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
    auto SC = [&]() -> const SceneView & { return *(const SceneView *)&ka->sc; };
    auto FP = [&]() -> const FrameParams & { return *(const FrameParams *)&ka->fp; };
    auto PA = [&]() -> const PoolParams & { return *(const PoolParams *)&ka->pp; };

    constexpr bool STATS = (FLAGS & 1) != 0, SUN = (FLAGS & 2) != 0, ALPHA = (FLAGS & 4) != 0, HBM = (FLAGS & 8) != 0;
    constexpr bool DEEP = (FLAGS & 32) != 0;         // deep tree: only the bottom levels of the traversal stacks are in LDS, the rest in HBM (PoolParams::stack_lds)
    constexpr bool EXT = (FLAGS & 16) != 0;          // the opt-in material model (drt.h drt_material_model): emissive term, mirror lobe, dielectric lobe
    constexpr uint32_t kWordBytes = HBM ? 16u : 4u, kStackEntryBytes = HBM ? 6u : 8u;
    extern __shared__ uint4 lds_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t wg = blockDim.x;
    const uint32_t P = PA().P;
    // (the parameters the hot phases use stay in scalar registers for the whole launch; everything else is read where it is used)
    const uint32_t k_ring_cap = PA().ring_cap;
    const uint32_t k_ring_shift = PA().ring_shift;
    const uint32_t k_min_fill = PA().min_fill;
    const uint32_t k_patience = PA().patience;
    const uint32_t k_n_loop = PA().n_loop;
    const uint32_t k_n_min_lanes = PA().n_min_lanes;
    const uint32_t k_n_fuse_loop = PA().n_fuse_loop;
    const uint32_t k_n_fuse_min = PA().n_fuse_min;
    const uint32_t k_dir_tries = PA().dir_tries;
    const uint32_t k_stack_entries = PA().stack_entries;
    const uint32_t k_stack_lds = DEEP ? PA().stack_lds : k_stack_entries;
    const uint32_t k_t_class0 = PA().t_class[0], k_t_class1 = PA().t_class[1], k_t_class2 = PA().t_class[2];
    const uint32_t ring_mask = k_ring_cap - 1u;
    const bool cold_lds = PA().cold_in_lds != 0;
    constexpr uint32_t kRings = SUN ? (uint32_t)kNQ : (uint32_t)kNQ - 1u;
    const PoolLayout lay = pool_layout(P, k_ring_cap, k_stack_lds, HBM ? 0u : pool_scene_bytes(SC()), cold_lds ? pool_cold_bytes(SC()) : 0u, kRings, kWordBytes, kStackEntryBytes, STATS ? kStatsBytes : 0u);
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(lds_raw);      // low 32 bits of the flat address = LDS offset
    const uint32_t ctrl = lds_base + lay.ctrl, rings = lds_base + lay.rings, stack = lds_base + lay.stack;
    const uint32_t qA = lds_base + lay.quads, qB = qA + P * 16u, qW = lds_base + lay.words;
    const uint32_t lds_inner = lds_base + lay.scene, lds_hot = lds_inner + SC().n_inner * 64u;
    const uint32_t lds_cold = lds_base + lay.cold, lds_mats = lds_cold + SC().n_tris * 32u, lds_texs = lds_mats + SC().n_mats * 16u;
    uint4 *const aux = PA().aux + (size_t)blockIdx.x * P;
    uint32_t *const aux_slot = PA().aux_slot + (size_t)blockIdx.x * P;
    float4 *const aux_light = PA().aux_light + (size_t)blockIdx.x * P;
    float4 *const aux_next = EXT ? PA().aux_next + (size_t)blockIdx.x * P * 2u : nullptr;

    if (FP().span && lane == 0) atomicMax(&FP().span[0], ~(unsigned long long)wall_clock64());

    // ---- prologue: scene copy, empty rings, every pool slot waits in E without a sample ----
    {
        const uint4 *g_inner = reinterpret_cast<const uint4 *>(SC().inner);
        const uint4 *g_hot = reinterpret_cast<const uint4 *>(SC().tri_hot);
        // (leaf references are rewritten while the records are staged: kLeafBit | count << 12 | first triangle -- reaching a leaf
        // then costs no LeafRange load; n_tris < 4096 is what path_pool_supports guarantees)
        for (uint32_t i = tid; !HBM && i < SC().n_inner * 4u; i += wg) {
            uint4 v = g_inner[i];
            if ((i & 3u) == 3u) {
                if (v.x & kLeafBit) { const LeafRange lr = SC().leaves[v.x & ~kLeafBit]; v.x = kLeafBit | ((uint32_t)lr.count << 12) | (uint32_t)lr.start; }
                if (v.y & kLeafBit) { const LeafRange lr = SC().leaves[v.y & ~kLeafBit]; v.y = kLeafBit | ((uint32_t)lr.count << 12) | (uint32_t)lr.start; }
            }
            st4(lds_inner + i * 16u, v);
        }
        for (uint32_t i = tid; !HBM && i < SC().n_tris * 3u; i += wg) st4(lds_hot + i * 16u, g_hot[i]);
        if (cold_lds) {
            const uint4 *g_cold = reinterpret_cast<const uint4 *>(SC().tri_cold);
            const uint4 *g_mats = reinterpret_cast<const uint4 *>(SC().mats);
            const uint4 *g_texs = reinterpret_cast<const uint4 *>(SC().texs);
            for (uint32_t i = tid; i < SC().n_tris * 2u; i += wg) st4(lds_cold + i * 16u, g_cold[i]);
            for (uint32_t i = tid; i < SC().n_mats; i += wg) st4(lds_mats + i * 16u, g_mats[i]);
            for (uint32_t i = tid; i < SC().n_texs; i += wg) st4(lds_texs + i * 16u, g_texs[i]);
        }
        for (uint32_t i = tid; i < kRings * k_ring_cap; i += wg) st_id(rings + i * 2u, kEmptyId);
        for (uint32_t i = tid; i < 32u + (STATS ? kStatsBytes / 4u : 0u); i += wg) st1(ctrl + i * 4u, 0u);
        for (uint32_t i = tid; i < P; i += wg) {                                                  // no sample yet
            if (HBM) st4(qW + i * 16u, make_uint4(0u, 0u, 0u, 0u));                                // {meta, hit triangle, leaf end, -}: all of it
            else st1(qW + i * 4u, kNoPrim);
        }
        __syncthreads();
        for (uint32_t i = tid; i < P; i += wg) st_id(rings + ((uint32_t)QE * k_ring_cap + i) * 2u, i);
        if (tid == 0) { st1(ctrl + QE * 8u + 4u, P); st1(ctrl + kCtrlLive, P); }
        __syncthreads();
    }

    // hbm-scene build (and every statistics build): an index that addresses global memory is range-checked first; a violation sets a
    // status bit (8..: the host reports DRT_ERR_DEVICE with it) and the access goes to element 0 -- a logic error must not become a
    // GPU memory fault, which can take the whole node down
    // (branch-free -- a branch between the loads of a step would serialise them: the index is clamped, the violation noted in a
    // register and reported once, after the batch)
    uint32_t violations = 0;
    // statistics build: the exact work counters of drt_counters (samples, rays, node visits, interior visits, triangle tests, textured /
    // flat hits, shadow rays, their interior visits and triangle tests) -- counted per lane where the reference's loops would
    // count them (oracle/drt_oracle.c), summed at the end; FrameParams::counters != nullptr asks for them
    enum : int { C_SAMPLES = 0, C_RAYS, C_NODES, C_INNER, C_TRIS, C_HTEX, C_HFLAT, C_SRAYS, C_SINNER, C_STRIS, C_TRIES, C_COUNT };
    uint32_t work[C_COUNT];
    for (int k = 0; k < C_COUNT; k++) work[k] = 0;
    auto count = [&](int what, uint32_t n = 1u) { if (STATS) work[what] += n; };
    auto checked = [&](uint32_t index, uint32_t limit, unsigned int code) -> uint32_t {
        if (!(HBM || STATS)) return index;
        violations |= index >= limit ? code : 0u;
        return min(index, limit - 1u);
    };
    // (the lds-scene production build: its traversal indices are 12-bit fields of LDS words and address LDS, which cannot fault;
    // the reporting checks -- even only those of the shading phases -- cost room 1.2 %.  What it reads from GLOBAL memory -- the
    // shading records of scenes whose cold data is not staged, the texels, the HBM part of the path state -- is addressed through
    // `held`: clamped, not reported; the path id itself is checked where it is claimed.)
    auto held = [&](uint32_t index, uint32_t limit, unsigned int code) -> uint32_t {
        return (HBM || STATS) ? checked(index, limit, code) : min(index, limit - 1u);
    };
    // ---- the path's words: meta (flags, bounce index; + hit triangle in the lds-scene build, + stack height in the hbm-scene build) ----
    auto meta_at = [&](uint32_t id) -> uint32_t { return qW + id * kWordBytes; };
    auto bounce_of = [&](uint32_t meta) -> uint32_t { return HBM ? (meta & 0xFFFFu) : ((meta >> 12) & (EXT ? 0x7FFFu : 0xFFFFu)); };      // (EXT: bit 27 is kMetal; path_pool_supports caps the bounce limit)
    auto prim_of = [&](uint32_t id, uint32_t meta) -> uint32_t { return HBM ? ld1(meta_at(id) + 4u) : (meta & 0xFFFu); };
    // (prim: only the lds-scene build keeps it in this word; the hbm-scene build's stays where T wrote it)
    auto make_meta = [&](uint32_t prim, uint32_t bounce, uint32_t flags) -> uint32_t { return HBM ? (bounce | flags) : (prim | (bounce << 12) | flags); };
    // what N and T leave behind: the leaf the path stands on (if any) and its stack height
    auto store_trav = [&](uint32_t id, int sp, uint32_t cur, uint32_t end, uint32_t meta) {
        if (HBM) {
            st1(qB + id * 16u + 12u, cur);
            st1(meta_at(id) + 8u, end);
            st1(meta_at(id), (meta & ~0x00FF0000u) | ((uint32_t)sp << 16));
        } else st1(qB + id * 16u + 12u, cur | (end << 12) | ((uint32_t)sp << 24));
    };
    auto fetch_tri = [&](int i) -> TriTest {
        const uint32_t q = lds_hot + __umul24((uint32_t)i, 48u);
        uint4 a, b; uint32_t c;
        if (HBM) {                                   // (i: range-checked by the caller, once per batch)
            const uint4 *g = reinterpret_cast<const uint4 *>(SC().tri_hot) + (size_t)i * 3;
            a = g[0]; b = g[1]; c = reinterpret_cast<const uint32_t *>(g)[8];
        } else { a = ld4(q); b = ld4(q + 16); c = ld1(q + 32); }
        TriTest t;
        t.v0 = mk3(u2f(a.x), u2f(a.y), u2f(a.z)); t.e1 = mk3(u2f(a.w), u2f(b.x), u2f(b.y)); t.e2 = mk3(u2f(b.z), u2f(b.w), u2f(c));
        return t;
    };
    // (lds-scene build) the triangle at LDS address q; q + 48 bytes past the last triangle is still inside the workgroup's LDS (pool_layout pads)
    auto fetch_tri_at = [&](uint32_t q) -> TriTest {
        const uint4 a = ld4(q), b = ld4(q + 16);
        const uint32_t c = ld1(q + 32);
        TriTest t;
        t.v0 = mk3(u2f(a.x), u2f(a.y), u2f(a.z)); t.e1 = mk3(u2f(a.w), u2f(b.x), u2f(b.y)); t.e2 = mk3(u2f(b.z), u2f(b.w), u2f(c));
        return t;
    };
    auto fetch_children = [&](uint32_t index) -> ChildPair {
        const uint32_t q = lds_inner + index * 64u;
        uint4 a, b, c; uint2 r;
        if (HBM) {
            index = checked(index, SC().n_inner, 0x200u);
            const uint4 *g = reinterpret_cast<const uint4 *>(SC().inner) + (size_t)index * 4;
            a = g[0]; b = g[1]; c = g[2]; r = *reinterpret_cast<const uint2 *>(g + 3);
        } else { a = ld4(q); b = ld4(q + 16); c = ld4(q + 32); r = ld2(q + 48); }
        ChildPair p;
        p.min1 = mk3(u2f(a.x), u2f(a.y), u2f(a.z)); p.max1 = mk3(u2f(a.w), u2f(b.x), u2f(b.y));
        p.min2 = mk3(u2f(b.z), u2f(b.w), u2f(c.x)); p.max2 = mk3(u2f(c.y), u2f(c.z), u2f(c.w));
        p.ref1 = r.x; p.ref2 = r.y;
        return p;
    };
    auto fetch_face_normal = [&](int prim) -> f3 {
        if (HBM) return ld3(SC().tri_hot[prim].fn);    // (prim: range-checked by the caller)
        const uint32_t q = lds_hot + __umul24((uint32_t)prim, 48u) + 36u;
        return mk3(u2f(ld1(q)), u2f(ld1(q + 4)), u2f(ld1(q + 8)));
    };
    auto fetch_cold = [&](int prim) -> TriCold {
        if (!cold_lds) return SC().tri_cold[held((uint32_t)prim, SC().n_tris, 0x80000u)];
        const uint32_t q = lds_cold + (uint32_t)prim * 32u;
        const uint4 a = ld4(q), b = ld4(q + 16);
        TriCold c;
        c.uv[0][0] = u2f(a.x); c.uv[0][1] = u2f(a.y); c.uv[1][0] = u2f(a.z); c.uv[1][1] = u2f(a.w);
        c.uv[2][0] = u2f(b.x); c.uv[2][1] = u2f(b.y); c.material = (int32_t)b.z; c._pad = 0;
        return c;
    };
    auto fetch_mat = [&](int index) -> MatDev {
        if (!cold_lds) return SC().mats[held((uint32_t)index, SC().n_mats, 0x2000u)];
        const uint4 a = ld4(lds_mats + (uint32_t)index * 16u);
        MatDev m; m.albedo[0] = u2f(a.x); m.albedo[1] = u2f(a.y); m.albedo[2] = u2f(a.z); m.tex = (int32_t)a.w;
        return m;
    };
    auto fetch_tex = [&](int index) -> TexDev {
        if (!cold_lds) return SC().texs[held((uint32_t)index, SC().n_texs, 0x4000u)];
        const uint4 a = ld4(lds_texs + (uint32_t)index * 16u);
        TexDev t; t.width = (int32_t)a.x; t.height = (int32_t)a.y; t.comps = (int32_t)a.z; t.offset = a.w;
        return t;
    };
    // AnyHit (AnyHit.cuh:8-28) through the readers above: the records come from LDS where they are staged, and in the hbm-scene
    // build the material and texture indices (read from global memory) are range-checked like every other index; same arithmetic
    // as device_access.hpp any_hit
    auto alpha_test = [&](int prim, f3 uvw) -> bool {
        const TriCold cold = fetch_cold(prim);                   // (prim: range-checked by the caller)
        const MatDev mat = fetch_mat(cold.material);
        if (mat.tex < 0) return true;
        const TexDev tex = fetch_tex(mat.tex);
        if (tex.comps < 4) return true;
        const float alpha = tex_get_alpha<true>(SC(), tex, interp_uv(cold, uvw));
        return !(alpha < 1);
    };
    // leaf -> its triangles [cur, end) and the T queue of its size class.  leaf_ref: count << 12 | first triangle in the lds-scene
    // build (the staged records carry it), the leaf's index in the hbm-scene build
    auto leaf_state = [&](uint32_t leaf_ref, uint32_t &cur, uint32_t &end) -> int {
        uint32_t count;
        if (HBM) { const LeafRange lr = SC().leaves[checked(leaf_ref, SC().n_leaves, 0x400u)]; cur = (uint32_t)lr.start; count = (uint32_t)lr.count; }
        else { cur = leaf_ref & 0xFFFu; count = leaf_ref >> 12; }
        end = cur + count;
        const uint32_t steps = (count + 1u) >> 1;
        return QT0 + (steps <= k_t_class0 ? 0 : (steps <= k_t_class1 ? 1 : (steps <= k_t_class2 ? 2 : 3)));
    };
    // traversal over: a path with a hit is shaded (B), one without ends on the sky (E)
    // (a shadow traversal: on to S, the second half of the shading)
    auto after_traversal = [&](float hit_t, bool shadow) -> int { return shadow ? QS : (hit_t < FLT_MAX ? QB : QE); };
    // Stack entries {node reference, slab distance}, [level][path]: 8 bytes in the lds-scene build; the hbm-scene build, whose deep trees
    // make the stacks most of a path's LDS, keeps the distances in one array and 16-bit references (bit 15 = leaf) in another:
    // 6 bytes, a quarter more paths per CU (path_pool_supports: fewer than 32768 interior nodes and leaves)
    // Deep trees: only the bottom k_stack_lds levels of every stack are in LDS; a path that goes deeper keeps the entries above
    // them in global memory (8 bytes each, [level][path] per workgroup) -- rare accesses, and the LDS they free holds more paths.
    const uint32_t stack_refs = stack + k_stack_lds * P * 4u;
    uint2 *const aux_stack = DEEP ? PA().aux_stack + (size_t)blockIdx.x * (k_stack_entries - k_stack_lds) * P : nullptr;
    auto stack_load = [&](int level, uint32_t id) -> uint2 {
        if (DEEP && __builtin_expect((uint32_t)level >= k_stack_lds, 0))
            return aux_stack[(size_t)(min((uint32_t)level, k_stack_entries - 1u) - k_stack_lds) * P + id];      // (clamped: a level beyond the tree's depth is a logic error, not a fault)
        const uint32_t at = (uint32_t)level * P + id;   // (LDS: cannot fault)
        if (!HBM) return ld2(stack + at * 8u);
        const uint32_t r16 = ld_u16(stack_refs + at * 2u);
        return make_uint2((r16 & 0x8000u) ? (kLeafBit | (r16 & 0x7FFFu)) : r16, ld1(stack + at * 4u));
    };
    auto stack_store = [&](int level, uint32_t id, uint2 e) {
        if (HBM || STATS) violations |= (uint32_t)level >= k_stack_entries ? 0x20000u : 0u;
        if (DEEP && __builtin_expect((uint32_t)level >= k_stack_lds, 0)) {
            aux_stack[(size_t)(min((uint32_t)level, k_stack_entries - 1u) - k_stack_lds) * P + id] = e;
            return;
        }
        const uint32_t at = (uint32_t)level * P + id;
        if (!HBM) { st2(stack + at * 8u, e); return; }
        st_u16(stack_refs + at * 2u, (e.x & kLeafBit) ? (0x8000u | (e.x & 0x7FFFu)) : e.x);
        st1(stack + at * 4u, e.y);
    };
    // One visit of BVHTraversal.cuh:33-72.  The entry to visit is `top` when have_top is set (an entry that would have been pushed
    // and popped again at once: it never goes through LDS), else the stack's top.  Returns the T queue when the path now stands on
    // a leaf ([cur, end) = its triangles), else -1 (entry culled, or an interior node: its far child went on
    // the stack, its near child -- the next visit -- into `top`).  (ray.dir is not used: the slab test needs origin and 1/dir.)
    auto pop_step = [&](const Ray &ray, float hit_t, int &sp, uint32_t id, uint32_t &cur, uint32_t &end, uint2 &top, bool &have_top, bool shadow) -> int {
        uint2 e = top;
        if (!have_top) { --sp; e = stack_load(sp, id); }
        have_top = false;
        int dest = -1;
        // :41 (without a hit, hit_t = FLT_MAX > dist); :38 was applied when the root was pushed
        if (!(hit_t < u2f(e.y))) {
            if (!shadow) count(C_NODES);                          // :43 (RayTest keeps no heat map)
            if (e.x & kLeafBit) dest = leaf_state(e.x & ~kLeafBit, cur, end);
            else {
                count(shadow ? C_SINNER : C_INNER);
                const ChildPair c = fetch_children(e.x);
                const float d1 = slab_entry_or_inf(c.min1, c.max1, ray);
                const float d2 = slab_entry_or_inf(c.min2, c.max2, ray);
                const bool first_is_1 = d1 > d2;          // farther child first; child 2 first on ties (:63-70)
                const uint32_t ra = first_is_1 ? c.ref1 : c.ref2, rb = first_is_1 ? c.ref2 : c.ref1;
                const float da = first_is_1 ? d1 : d2, db = first_is_1 ? d2 : d1;
                if (da < hit_t) { stack_store(sp, id, make_uint2(ra, f2u(da))); ++sp; }
                if (db < hit_t) {
                    // the near child is this path's next visit and passes :41 (nothing changes hit_t in between): a leaf goes
                    // straight to T, an interior node stays in registers
                    if (rb & kLeafBit) { dest = leaf_state(rb & ~kLeafBit, cur, end); if (!shadow) count(C_NODES); }     // (its pop, which passes :41)
                    else { top = make_uint2(rb, f2u(db)); have_top = true; }
                }
            }
        }
        return dest;
    };
    // a path leaves a batch with an entry still in registers: it goes on the stack after all
    auto spill_top = [&](int &sp, uint32_t id, const uint2 &top, bool &have_top) {
        if (have_top) { stack_store(sp, id, top); ++sp; have_top = false; }
    };
    // Push every lane's path id (dest >= 0) to its destination queue: one atomic add per destination present in the wave
    // (issued together by the first lane of each group), then the ids go to consecutive ring positions.
    auto push_group = [&](int dest, uint32_t id) {
        unsigned long long todo = pp_ballot(dest >= 0);
        uint32_t add_count = 0, my_rank = 0;
        int my_leader = 0;
        while (todo) {
            const int first = __builtin_ctzll(todo);
            const int d = __builtin_amdgcn_readlane(dest, first);
            const unsigned long long m = pp_ballot(dest == d);
            if (dest == d) { my_rank = (uint32_t)pp_rank(m); my_leader = first; add_count = (uint32_t)__popcll(m); }
            todo &= ~m;
        }
        uint32_t base = 0;
        if (dest >= 0 && lane == my_leader) base = lds_add(ctrl + (uint32_t)dest * 8u + 4u, add_count);
        base = (uint32_t)__builtin_amdgcn_ds_bpermute(my_leader << 2, (int)base);
        if (dest >= 0) {
            // The slot may still hold the entry of the previous lap: its consumer has claimed it (at most P <= ring_cap ids are ever
            // outstanding, so the head is past it) but may not have read and cleared it yet -- at the start, when E's ring holds
            // all P ids, a path that goes straight back to E lands on exactly such a slot.  Wait for the slot to be empty.
            const uint32_t pos = base + my_rank;
            const uint32_t at = rings + ((uint32_t)dest * k_ring_cap + (pos & ring_mask)) * 2u;
            if (__builtin_expect(ld_id(at) != kEmptyId, 0)) {             // (rare: kept out of the straight path)
                uint32_t spins = 0;
#pragma nounroll
                while (ld_id(at) != kEmptyId)
                    if (++spins > (1u << 24)) { st1_shared(ctrl + kCtrlAbort, 3u); if (PA().status) atomicOr(PA().status, 4u); break; }
            }
            st_id(at, id | (((pos >> k_ring_shift) & 15u) << 12));       // the entry carries the lap of its position
        }
    };
    const f3 root_min = ld3(SC().root_min), root_max = ld3(SC().root_max);
    // TraceRay.cu:15-20 + BVHTraversal.cuh:22-26,38: the root goes on the stack with its slab distance when -1 < d < FLT_MAX
    uint32_t root_ref = SC().root_ref;                      // (a one-leaf scene: the same self-describing form as the staged records)
    if (!HBM && root_ref != kNoNode && (root_ref & kLeafBit)) { const LeafRange lr = SC().leaves[root_ref & ~kLeafBit]; root_ref = kLeafBit | ((uint32_t)lr.count << 12) | (uint32_t)lr.start; }
    auto begin_closest = [&](const Ray &ray, uint2 &top) -> bool {
        if (SC().root_ref == kNoNode) return false;
        const float d = slab_intersect(root_min, root_max, ray);
        if (!(-1.0f < d && d < FLT_MAX)) return false;
        top = make_uint2(root_ref, f2u(d));
        return true;
    };
    // Traversal registers of a lane: what the N loop works on.  A lane gets them either from LDS (an N batch: a path whose
    // traversal is under way) or from a launch (B / S / R / E start a new ray: the root's entry is its first visit) -- the
    // launching batch then runs the N loop itself, so that a new ray's first pops cost no trip through the N queue.
    struct Trav {
        Ray ray; float hit_t; int sp; uint32_t cur, end, meta; uint2 top; bool have_top, shadow;
        bool on;            // takes part in the N loop of this batch
        bool fresh;         // launched by this batch: its whole state is stored afterwards, not just the traversal part
        uint32_t hit_store; // fresh: the hit distance written to LDS (FLT_MAX; 0 for a path that never traces, RayGen.cuh:88)
    };
    auto trav_idle = [&]() -> Trav {
        Trav t; t.ray = make_ray(mk3(0, 0, 0), mk3(0, 0, 1)); t.hit_t = FLT_MAX; t.sp = 0; t.cur = t.end = t.meta = 0; t.top = make_uint2(0u, 0u);
        t.have_top = t.shadow = t.on = t.fresh = false; t.hit_store = f2u(FLT_MAX);
        return t;
    };
    // A new ray (TraceRay.cu:15-20): hit distance FLT_MAX -- for a shadow ray it stays there, so that N's culls
    // (BVHTraversal.cuh:41, :63-70) never apply: RayTest has none
    auto launch_with = [&](Trav &t, const Ray &ray, uint32_t w_word, uint2 top, bool have_top, bool shadow) {
        t.ray = ray; t.hit_t = FLT_MAX; t.sp = 0; t.cur = t.end = 0; t.meta = w_word; t.top = top; t.have_top = have_top; t.shadow = shadow;
        t.on = have_top; t.fresh = true; t.hit_store = f2u(FLT_MAX);
    };
    auto launch_ray = [&](Trav &t, const Ray &ray, uint32_t bounce, bool trace) {
        uint2 top = make_uint2(0u, 0u);
        const bool have_top = trace && begin_closest(ray, top);
        if (trace) count(C_RAYS);                        // TraceRay.cu:15
        launch_with(t, ray, make_meta(kNoPrim, bounce, kHasSample), top, have_top, false);
        if (!trace) t.hit_store = 0u;                 // RayGen.cuh:88: the loop body never runs, the sample is black (E adds no sky light)
    };
    // RayTest (BVHTraversal.cuh:76-134): the root is visited unless its slab test says "behind" (:95-103), no distance culls
    auto launch_shadow = [&](Trav &t, const Ray &ray, uint32_t w_word) {
        const bool have_top = SC().root_ref != kNoNode && !(slab_intersect(root_min, root_max, ray) < 0);
        count(C_SRAYS);
        launch_with(t, ray, w_word, make_uint2(root_ref, f2u(0.0f)), have_top, true);
    };

    const int wave = tid >> 6;
    uint32_t polls = 0, idle_polls = 0;
    uint32_t my_shard = (uint32_t)__builtin_amdgcn_readfirstlane((int)blockIdx.x) & (kSampleShards - 1u);      // scalar: the sample counter this wave draws from
    uint32_t rot = (uint32_t)__builtin_amdgcn_readfirstlane(wave) % (uint32_t)kNQ;      // scalar: where this wave's round robin over the queues stands
    unsigned long long s_work[6] = { 0, 0, 0, 0, 0, 0 };      // STATS: N iterations / lane pops, T steps / lane steps, direction-try iterations / lane tries
    // (the per-queue sums {batches, lanes, ticks} are kept in LDS, kCtrlStats: indexed by the queue, they would otherwise be arrays in scratch
    // memory and the statistics build a different kernel from the one it describes)
    unsigned long long s_claim = 0, s_idle = 0, s_lost = 0, s_fail = 0, s_fail_ticks = 0, s_idle_ticks = 0;
    const unsigned long long s_t_start = STATS ? __builtin_amdgcn_s_memtime() : 0;
    unsigned long long s_t0 = s_t_start;
    for (;;) {
        asm volatile("" : "+s"(ka));        // a fresh view of the arguments: nothing read through it is carried over from the batch before
        if (STATS) s_t0 = __builtin_amdgcn_s_memtime();
        // (lds-scene build: a wave that is claiming or pushing goes ahead of the waves inside a phase -- these are short chains of LDS
        // round trips, and the sooner they are through the sooner 64 more lanes have work: room -2 %, cornell -1 %; the hbm-scene
        // build, which waits for L2 everywhere, loses 1 % with it)
        if (!HBM) __builtin_amdgcn_s_setprio(1);
        // ---------------- choose a queue and claim up to 64 of its ids ----------------
        // Queues holding a full batch are shared out round robin (the waves of a workgroup would otherwise all race for
        // the same one); with none, the fullest queue is taken -- after a short wait for company unless the launch is draining.
        // One LDS read fetches every control word: lanes 0..8 their queue's {head, tail}, lane 10 {live, abort}, lane 11 {exhausted}.
        uint2 my_ctrl = make_uint2(0u, 0u);
        if (lane < 12) my_ctrl = ld2_shared(ctrl + (uint32_t)lane * 8u);
        const int my_avail = lane < kNQ ? (int)(my_ctrl.y - my_ctrl.x) : 0;
        const uint32_t live = (uint32_t)__builtin_amdgcn_readlane((int)my_ctrl.x, 10), aborted = (uint32_t)__builtin_amdgcn_readlane((int)my_ctrl.y, 10);
        const uint32_t exhausted = (uint32_t)__builtin_amdgcn_readlane((int)my_ctrl.x, 11);
        int q = -1, avail = 0;
        const unsigned full_mask = (unsigned)pp_ballot(my_avail >= 64) & ((1u << kNQ) - 1u);
        if (full_mask) {
            // the first queue with a full batch at or after `rot` (scalar arithmetic: rot stays in [0, kNQ))
            const unsigned turned = ((full_mask >> rot) | (full_mask << ((uint32_t)kNQ - rot))) & ((1u << kNQ) - 1u);
            const int k = __builtin_ctz(turned) + (int)rot;
            q = k >= kNQ ? k - kNQ : k;
            avail = 64;
        } else {
#pragma unroll
            for (int k = 0; k < kNQ; k++) {
                const int a = __builtin_amdgcn_readlane(my_avail, k);
                if (a > avail) { avail = a; q = k; }
            }
        }
        if (aborted != 0) break;
        if (q < 0) {
            if (live == 0) break;                                         // every pool slot retired: the launch is done
            __builtin_amdgcn_s_sleep(8);
            if (STATS) { s_idle++; s_idle_ticks += __builtin_amdgcn_s_memtime() - s_t0; }
            if (++idle_polls > (1u << 22)) { st1_shared(ctrl + kCtrlAbort, 1u); if (lane == 0 && PA().status) atomicOr(PA().status, 1u); break; }
            continue;
        }
        if ((uint32_t)avail < k_min_fill && polls < k_patience && exhausted == 0) {
            ++polls;
            __builtin_amdgcn_s_sleep(2);
            if (STATS) { s_idle++; s_idle_ticks += __builtin_amdgcn_s_memtime() - s_t0; }
            continue;
        }
        uint32_t head = (uint32_t)__builtin_amdgcn_readlane((int)my_ctrl.x, q);
        const uint32_t tail = (uint32_t)__builtin_amdgcn_readlane((int)my_ctrl.y, q);
        uint32_t n = (uint32_t)min(avail, 64);
        int won = 0;
        // compare-and-swap on the head; a wave that loses the race reads the queue's head and tail again (one LDS load, every lane
        // the same address) and tries for what is there now -- cheaper than choosing a queue afresh
        uint32_t tail_now = tail;
        for (int attempt = 0; attempt < 6 && !won; attempt++) {
            uint32_t seen = head;
            if (lane == 0) won = lds_cas_seen(ctrl + (uint32_t)q * 8u, head, head + n, seen) ? 1 : 0;
            won = __builtin_amdgcn_readfirstlane(won);
            if (won) break;
            if (STATS) s_lost++;
            const uint2 ht = ld2_shared(ctrl + (uint32_t)q * 8u);
            head = (uint32_t)__builtin_amdgcn_readfirstlane((int)ht.x);
            tail_now = (uint32_t)__builtin_amdgcn_readfirstlane((int)ht.y);
            const int left = (int)(tail_now - head);
            if (left < (int)k_min_fill && exhausted == 0) break;
            if (left <= 0) break;
            n = (uint32_t)min(left, 64);
        }
        (void)tail_now;
        rot = rot + 1u >= (uint32_t)kNQ ? 0u : rot + 1u;
        if (!won) { if (STATS) { s_fail++; s_fail_ticks += __builtin_amdgcn_s_memtime() - s_t0; } continue; }   // the queue went to other waves: look again
        polls = 0; idle_polls = 0;
        const bool active = (uint32_t)lane < n;
        uint32_t id = 0;
        if (active) {
            // only the entry written for THIS position counts: empty = its producer has reserved the position and not written yet;
            // another lap's tag = the entry of the position one lap earlier, which its own consumer (a wave that has claimed it and
            // not got round to reading it) is still to take -- taking that one would put a path in two hands
            const uint32_t pos = head + (uint32_t)lane;
            const uint32_t at = rings + ((uint32_t)q * k_ring_cap + (pos & ring_mask)) * 2u;
            const uint32_t my_lap = (pos >> k_ring_shift) & 15u;
            uint32_t e = ld_id(at);
            if (__builtin_expect(e == kEmptyId || (e >> 12) != my_lap, 0)) {          // (rare: kept out of the straight path)
                uint32_t spins = 0;
#pragma nounroll
                for (;;) {
                    e = ld_id(at);
                    if (e != kEmptyId && (e >> 12) == my_lap) break;
                    if (++spins > (1u << 24)) { st1_shared(ctrl + kCtrlAbort, 2u); if (PA().status) atomicOr(PA().status, 2u); e = 0; break; }
                }
            }
            id = e & kIdMask;
            st_id(at, kEmptyId);
            // every per-path address below (LDS and the HBM part of the state: aux, aux_slot, aux_light) is formed from this id
            if (id >= P) { st1_shared(ctrl + kCtrlAbort, 4u); if (PA().status) atomicOr(PA().status, 0x40000u); id = 0; }
        }
        lds_acquire();
        unsigned long long s_t1 = 0;
        if (STATS) { s_t1 = __builtin_amdgcn_s_memtime(); s_claim += s_t1 - s_t0; if (lane == 0) { lds_add64(ctrl + kCtrlStats + (uint32_t)q * 24u, 1ull); lds_add64(ctrl + kCtrlStats + (uint32_t)q * 24u + 8u, (unsigned long long)n); } }
        if (!HBM) __builtin_amdgcn_s_setprio(0);
        int dest = -1;                                                    // queue this lane's path goes to next
        Trav tr = trav_idle();                                            // this lane's traversal registers (an N batch loads them, a launch sets them)
        // bounce direction being drawn (B, R): randomUnitSphereVec3 is a rejection loop (Random.cu:50-58, ~2.9 candidates on
        // average, a long tail); a batch draws at most k_dir_tries candidates per path with every lane that still needs one,
        // then launches the rays that have their direction and sends the others to R with their RNG state
        bool need_dir = false;
        f3 dir_origin = mk3(0, 0, 0), dir_normal = mk3(0, 0, 0), dir_thr = mk3(1, 1, 1);
        uint32_t dir_seed = 0, dir_bounce = 0, dir_tries = 0;
        // material model (EXT): the mirror lobe draws the same candidates; the ray is reflect(v, N) [in dir_normal] + roughness * candidate,
        // and the path ends if that points into the surface (below the shading normal: +- the face normal of dir_prim, found again when needed --
        // carrying it through the candidate loop cost the three registers that kept <16> at one workgroup per CU)
        bool dir_metal = false, dir_back = false;
        float dir_rough = 0.0f;
        uint32_t dir_prim = kNoPrim;
        auto draw_and_launch = [&](bool store_throughput) {
            f3 p = mk3(0, 0, 0);
            bool have = false;
            for (uint32_t k = 0; k < k_dir_tries; k++) {
                const bool go = need_dir && !have;
                if (pp_ballot(go) == 0) break;
                if (STATS) { s_work[4]++; s_work[5] += (unsigned long long)__popcll(pp_ballot(go)); }
                // (the cycle guard of device_math.hpp: the kMaxTries-th candidate is taken whatever it is)
                if (go) { count(C_TRIES); have = random_unit_sphere_try(dir_seed, p) || ++dir_tries >= (uint32_t)kMaxTries; }
            }
            if (need_dir) {
                if (have) {
                    // the RNG state goes on with the path (:91 of the next bounce reads it); B also has a new throughput
                    if (store_throughput) aux[id] = make_uint4(f2u(dir_thr.x), f2u(dir_thr.y), f2u(dir_thr.z), dir_seed);
                    else reinterpret_cast<uint32_t *>(aux + id)[3] = dir_seed;
                    const f3 d = (EXT && dir_metal) ? dir_normal + p * dir_rough : dir_normal + p;       // :134 / the mirror lobe
                    bool absorbed = false;
                    if (EXT && dir_metal) {
                        const f3 fn = fetch_face_normal((int)dir_prim);
                        absorbed = !(dot(d, dir_back ? (-1.f * fn) : fn) > 0.0f);
                    }
                    if (absorbed) {
                        // scattered into the surface: absorbed -- the path ends here without reaching the sky
                        st1(qA + id * 16u + 12u, 0u);
                        st1(meta_at(id), make_meta(kNoPrim, 0u, kHasSample));
                        dest = QE;
                    } else launch_ray(tr, make_ray(dir_origin, d), dir_bounce, true);
                } else {
                    if (store_throughput) aux[id] = make_uint4(f2u(dir_thr.x), f2u(dir_thr.y), f2u(dir_thr.z), dir_seed);
                    st4(qA + id * 16u, make_uint4(f2u(dir_origin.x), f2u(dir_origin.y), f2u(dir_origin.z), dir_seed));
                    st4(qB + id * 16u, make_uint4(f2u(dir_normal.x), f2u(dir_normal.y), f2u(dir_normal.z), dir_tries));
                    if (EXT && dir_metal) st1(meta_at(id), make_meta(dir_prim, dir_bounce, kHasSample | kMetal | (dir_back ? kBackFace : 0u)));   // (hbm-scene: the hit triangle stays in word 1)
                    else st1(meta_at(id), make_meta(kNoPrim, dir_bounce, kHasSample));
                    dest = QR;
                }
            }
        };

        if (q == QN) {
            // ============ N: pop stack entries until the path stands on a leaf (BVHTraversal.cuh:33-72) ============
            // (the loop itself is below, shared with the batches that launch rays)
            if (active) {
                const uint4 A = ld4(qA + id * 16u), B = ld4(qB + id * 16u);
                if (HBM || SUN) tr.meta = ld1(meta_at(id));
                tr.sp = HBM ? (int)((tr.meta >> 16) & 0xFFu) : (int)(B.w >> 24);
                tr.hit_t = u2f(A.w);
                tr.ray = make_ray(mk3(u2f(A.x), u2f(A.y), u2f(A.z)), mk3(u2f(B.x), u2f(B.y), u2f(B.z)));     // 1/dir again (Ray.cuh:7-9): 12 bytes of LDS per path saved
                if (SUN) tr.shadow = (tr.meta & kShadow) != 0;
                tr.on = true;
            }
        } else if (q >= QT0 && q < QB) {
            // ============ T: the triangles of one leaf, two per step (Intersection.cu:4-36, BVHTraversal.cuh:46-57) ============
            Ray ray = make_ray(mk3(0, 0, 0), mk3(0, 0, 1));
            float hit_t = FLT_MAX;
            uint32_t hit_prim = 0, meta = 0;
            int cur = 0, end = 0, sp = 0;
            bool shadow = false, occluded = false;
            if (active) {
                const uint4 A = ld4(qA + id * 16u), B = ld4(qB + id * 16u);
                ray.orig = mk3(u2f(A.x), u2f(A.y), u2f(A.z)); hit_t = u2f(A.w);
                ray.dir = mk3(u2f(B.x), u2f(B.y), u2f(B.z));
                if (HBM || SUN) meta = ld1(meta_at(id));
                if (HBM) {
                    end = (int)(checked(ld1(meta_at(id) + 8u) - 1u, SC().n_tris, 0x100u) + 1u);         // [cur, end) within the triangles
                    cur = (int)min(B.w, (uint32_t)end);
                    sp = (int)((meta >> 16) & 0xFFu);
                }
                else { cur = (int)(B.w & 0xFFFu); end = (int)((B.w >> 12) & 0xFFFu); sp = (int)(B.w >> 24); }
                if (SUN) shadow = (meta & kShadow) != 0;
            }
            const float hit_t_in = hit_t;
            // hbm-scene: four triangles per step in the wide build -- four loads in flight per lane instead of two (the step waits for
            // memory, not for the arithmetic) --, two in the narrow one (pool_narrow); tested and applied in the leaf's order
            while (HBM && pp_ballot(cur < end) != 0) {
                if (cur < end) {
                    constexpr int kWide = pool_narrow(FLAGS) ? 2 : 4;
                    int idx[kWide];
                    TriTest tri[kWide];
#pragma unroll
                    for (int k = 0; k < kWide; k++) { idx[k] = min(cur + k, end - 1); tri[k] = fetch_tri(idx[k]); }
                    const int first = cur;
                    cur = min(cur + kWide, end);
#pragma unroll
                    for (int k = 0; k < kWide; k++) {
                        float t, u, v;
                        const bool h = tri_intersect_flat(ray, tri[k].v0, tri[k].e1, tri[k].e2, t, u, v) & (first + k < end);
                        if (SUN && shadow) {
                            if (STATS && first + k < end && !occluded) count(C_STRIS);
                            if (h && !occluded && (!ALPHA || alpha_test(idx[k], mk3(1.0f - u - v, u, v)))) { occluded = true; cur = end; sp = 0; }
                        } else if (STATS && first + k < end) count(C_TRIS);
                        if (SUN && shadow) {
                        } else if (h && t < hit_t && (!ALPHA || alpha_test(idx[k], mk3(1.0f - u - v, u, v)))) { hit_t = t; hit_prim = (uint32_t)idx[k]; }
                    }
                }
            }
            if (!HBM) {
                // One LDS address per lane walks the leaf: a pair is at [addr, addr + 96), the second triangle at the constant
                // offset 48.  A lane whose leaf is used up stays where it is and tests its last pair again, result ignored: no
                // exec-mask bookkeeping and no loop-carried copies inside the loop (the lanes of a batch are on leaves of one size
                // class, so they nearly all run to the end together anyway).
                int left = end - cur;                                     // triangles still to test (<= 0: through)
                uint32_t addr = lds_hot + __umul24((uint32_t)cur, 48u);
                for (;;) {
                    const bool go = left > 0;
                    if (pp_ballot(go) == 0) break;
                    if (STATS) { s_work[2]++; s_work[3] += (unsigned long long)__popcll(pp_ballot(go)) + (unsigned long long)__popcll(pp_ballot(left > 1)); }    // (lane steps in half steps: triangles tested)
                    const bool two = left > 1;
                    // (fetching the next pair while this one is tested was measured: 80 VGPRs, cornell -4 %, room -2.6 %)
                    const TriTest ta = fetch_tri_at(addr), tb = fetch_tri_at(addr + 48u);
                    float t0, u0, v0, t1, u1, v1;
                    const bool h0 = tri_intersect_flat(ray, ta.v0, ta.e1, ta.e2, t0, u0, v0) & go;
                    const bool h1 = tri_intersect_flat(ray, tb.v0, tb.e1, tb.e2, t1, u1, v1) & two;
                    const int i = cur, j = cur + 1;
                    if (SUN && shadow) {                                  // RayTest: any accepted hit ends the traversal (:105-117)
                        const bool occ0 = h0 && (!ALPHA || alpha_test(i, mk3(1.0f - u0 - v0, u0, v0)));
                        const bool occ = occ0 || (h1 && (!ALPHA || alpha_test(j, mk3(1.0f - u1 - v1, u1, v1))));
                        if (STATS && go) count(C_STRIS, occ0 || !two ? 1u : 2u);          // (the reference stops at the first accepted hit)
                        if (occ) { occluded = true; left = 0; sp = 0; }
                    } else {
                        if (STATS && go) count(C_TRIS, two ? 2u : 1u);
                        // (the barycentrics are not kept: B computes them again for the one triangle that wins)
                        if (h0 && t0 < hit_t && (!ALPHA || alpha_test(i, mk3(1.0f - u0 - v0, u0, v0)))) { hit_t = t0; hit_prim = (uint32_t)i; }
                        if (h1 && t1 < hit_t && (!ALPHA || alpha_test(j, mk3(1.0f - u1 - v1, u1, v1)))) { hit_t = t1; hit_prim = (uint32_t)j; }
                    }
                    const int step = go ? 2 : 0;
                    cur += step; left -= step; addr += (uint32_t)step * 48u;
                }
            }
            if (active) {
                if (SUN && occluded) { meta |= kOccluded; if (!HBM) st1(meta_at(id), meta); }      // (hbm-scene: store_trav writes meta)
                if (hit_t < hit_t_in) {
                    st1(qA + id * 16u + 12u, f2u(hit_t));
                    if (HBM) st1(meta_at(id) + 4u, hit_prim);
                    else st1(meta_at(id), (ld1(meta_at(id)) & ~0xFFFu) | hit_prim);
                }
                // What the next pops would do while the top of the stack is culled (:41) or a leaf: done here, the path goes
                // straight to its next leaf; an interior node is left to N.
                uint32_t next_cur = 0, next_end = 0;
                while (sp > 0) {
                    const uint2 e = stack_load(sp - 1, id);
                    if (hit_t < u2f(e.y)) { --sp; continue; }
                    if (e.x & kLeafBit) { --sp; dest = leaf_state(e.x & ~kLeafBit, next_cur, next_end); if (!(SUN && shadow)) count(C_NODES); }
                    break;
                }
                if (dest < 0) dest = sp > 0 ? QN : after_traversal(hit_t, shadow);
                store_trav(id, sp, next_cur, next_end, meta);
            }
        } else if (q == QR) {
            // ============ R: more candidates for directions B could not settle (Random.cu:50-58), then the launch ============
            if (active) {
                const uint4 A = ld4(qA + id * 16u), B = ld4(qB + id * 16u);
                need_dir = true;
                dir_origin = mk3(u2f(A.x), u2f(A.y), u2f(A.z)); dir_seed = A.w;
                dir_normal = mk3(u2f(B.x), u2f(B.y), u2f(B.z)); dir_tries = B.w;
                const uint32_t W = ld1(meta_at(id));
                dir_bounce = bounce_of(W);
                if (EXT && (W & kMetal)) {                     // the mirror lobe: roughness and shading normal again from the hit triangle
                    dir_prim = held(prim_of(id, W), SC().n_tris, 0x800u);
                    dir_back = (W & kBackFace) != 0;
                    dir_rough = SC().mats_ext[held((uint32_t)fetch_cold((int)dir_prim).material, SC().n_mats, 0x2000u)].roughness;
                    dir_metal = true;
                }
            }
            draw_and_launch(false);
        } else if (q == QB) {
            // ============ B: shade the hit, draw the bounce direction, launch the bounce ray (RayGen.cuh:90-134) ============
            if (active) {
                const uint4 E = aux[id];                                                   // {throughput, seed}
                const uint4 A = ld4(qA + id * 16u), B = ld4(qB + id * 16u);
                const uint32_t W = ld1(meta_at(id));
                Ray ray; ray.orig = mk3(u2f(A.x), u2f(A.y), u2f(A.z)); ray.dir = mk3(u2f(B.x), u2f(B.y), u2f(B.z)); ray.inv_dir = ray.dir;
                const float hit_t = u2f(A.w);
                const int hit_prim = (int)checked(prim_of(id, W), SC().n_tris, 0x800u);
                uint32_t bounce = bounce_of(W);
                // the barycentrics of the hit: the winning triangle's test once more (same inputs, same bits as in T)
                float hit_u, hit_v, t_again;
                const TriTest tri = fetch_tri(hit_prim);
                (void)tri_intersect_flat(ray, tri.v0, tri.e1, tri.e2, t_again, hit_u, hit_v);
                uint32_t seed = E.w + bounce;                                              // :91
                f3 throughput = mk3(u2f(E.x), u2f(E.y), u2f(E.z));
                const f3 uvw = mk3(1.0f - hit_u - hit_v, hit_u, hit_v);                    // Intersection.cu:31
                f3 position, normal;                                                       // ClosestHit.cuh:13-24
                const bool front_face = closest_hit_frame(ray, hit_t, fetch_face_normal(hit_prim), position, normal);
                const TriCold cold = fetch_cold(hit_prim);                                 // :111-118
                const MatDev mat = fetch_mat(cold.material);
                count(mat.tex < 0 ? C_HFLAT : C_HTEX);
                // ---- opt-in material model (EXT builds; drt.h drt_material_model -- this library's rule, stated first in oracle/drt_oracle.c) ----
                MatExt ext;
                ext.emissive[0] = ext.emissive[1] = ext.emissive[2] = 0; ext.roughness = 0; ext.metallic = 0; ext.transmission = 0; ext.refractive_index = 1;
                if (EXT) ext = SC().mats_ext[held((uint32_t)cold.material, SC().n_mats, 0x2000u)];
                if (EXT && FP().ext_emissive) {              // emission seen through the path so far, before this hit's albedo
                    float4 *const lp = aux_light + id;
                    const f3 light = mk3(lp->x, lp->y, lp->z) + (ld3(ext.emissive) * FP().ext_emissive_scale) * throughput;
                    *lp = make_float4(light.x, light.y, light.z, 0.0f);
                }
                if (mat.tex < 0) throughput = throughput * ld3(mat.albedo);
                else throughput = throughput * tex_get_pixel<true>(SC(), fetch_tex(mat.tex), interp_uv(cold, uvw));
                const f3 origin = position + (normal * 0.001f);                            // :121
                const bool glass = EXT && FP().ext_transmission && ext.transmission != 0;
                const bool mirror = EXT && !glass && FP().ext_specular && ext.metallic != 0;
                f3 refl = mk3(0, 0, 0);
                if (EXT && (glass || mirror)) { const f3 v = normalize(ray.dir); refl = v - normal * (2.0f * dot(v, normal)); }
                // the dielectric interface (Random.cu:26-40): the continuation ray is decided here, with one randomFloat -- after the
                // sun's draws, which come first in the reference's iteration (:124-134)
                auto dielectric = [&](uint32_t &rng, f3 &o2, f3 &d2) {
                    const f3 v = normalize(ray.dir);
                    const float cos_theta = fminf(dot(mk3(v.x * -1.f, v.y * -1.f, v.z * -1.f), normal), 1.0f);      // Random.cu:28
                    const float ri = front_face ? 1.0f / ext.refractive_index : ext.refractive_index;
                    const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
                    bool reflect = ri * sin_theta > 1.0f;                                                        // total internal reflection
                    float r0 = (1 - ri) / (1 + ri);                                                              // :37-39, pow(x, 5) = ((x x)(x x)) x
                    r0 = r0 * r0;
                    const float om = 1 - cos_theta;
                    const float schlick = r0 + (1 - r0) * (((om * om) * (om * om)) * om);
                    if (!reflect) reflect = schlick > random_float(rng);
                    if (reflect) { o2 = origin; d2 = refl; }
                    else {
                        const f3 perp = (v + normal * cos_theta) * ri;                                           // :29
                        const f3 par = normal * (-sqrtf(fabsf(1.0f - dot(perp, perp))));                         // :30
                        o2 = position - (normal * 0.001f); d2 = perp + par;
                    }
                };
                if (SUN) {
                    // :124-128: the sun's shadow ray starts where the bounce ray will (its origin waits in A), the normal is found
                    // again from the triangle and the side bit; the rest of this iteration is S's, after the shadow traversal
                    const Ray sun_ray = make_ray(origin, ld3(FP().sunpos) + random_unit_vec3(seed) * 1.5f);
                    if (EXT) {
                        // what S will need and the shadow ray overwrites: tag 2 = the dielectric's ray, decided now; 1 = mirror lobe
                        // {reflect(v, N), roughness}; 0 = the reference's diffuse bounce
                        f3 o2 = origin, d2 = refl;
                        if (glass) dielectric(seed, o2, d2);
                        aux_next[2u * id] = make_float4(d2.x, d2.y, d2.z, glass ? 2.0f : (mirror ? 1.0f : 0.0f));
                        aux_next[2u * id + 1u] = glass ? make_float4(o2.x, o2.y, o2.z, 0.0f) : make_float4(ext.roughness, 0.0f, 0.0f, 0.0f);
                    }
                    aux[id] = make_uint4(f2u(throughput.x), f2u(throughput.y), f2u(throughput.z), seed);
                    launch_shadow(tr, sun_ray, make_meta((uint32_t)hit_prim, bounce, kHasSample | kShadow | (front_face ? 0u : kBackFace)));
                } else {
                    ++bounce;
                    if ((int)bounce <= FP().bounce_limit && glass) {
                        f3 o2, d2;
                        dielectric(seed, o2, d2);
                        aux[id] = make_uint4(f2u(throughput.x), f2u(throughput.y), f2u(throughput.z), seed);
                        launch_ray(tr, make_ray(o2, d2), bounce, true);
                    } else if ((int)bounce <= FP().bounce_limit) {                           // :88 loop condition
                        need_dir = true; dir_origin = origin; dir_normal = mirror ? refl : normal; dir_seed = seed; dir_bounce = bounce; dir_tries = 0;
                        dir_thr = throughput;
                        if (EXT) { dir_metal = mirror; dir_rough = ext.roughness; dir_prim = (uint32_t)hit_prim; dir_back = !front_face; }
                    } else {
                        st1(qA + id * 16u + 12u, 0u);    // the path ends without reaching the sky: E adds no light (hit_t != FLT_MAX)
                        dest = QE;
                    }
                }
            }
            if (!SUN) draw_and_launch(true);
        } else if (SUN && q == QS) {
            // ============ S: the shadow traversal is over: sunlight, then the bounce direction and ray (RayGen.cuh:126-134) ============
            if (active) {
                const uint4 E = aux[id];
                const uint4 A = ld4(qA + id * 16u);
                const uint32_t W = ld1(meta_at(id));
                if (!(W & kOccluded)) {
                    float4 *const lp = aux_light + id;
                    const f3 light = mk3(lp->x, lp->y, lp->z) + ld3(FP().suncol) * mk3(u2f(E.x), u2f(E.y), u2f(E.z));     // :126-127
                    *lp = make_float4(light.x, light.y, light.z, 0.0f);
                }
                const uint32_t bounce = bounce_of(W) + 1u;
                float4 next0 = make_float4(0, 0, 0, 0);
                if (EXT) next0 = aux_next[2u * id];                                         // what B decided (tag in .w)
                if ((int)bounce <= FP().bounce_limit && EXT && next0.w == 2.0f) {
                    const float4 o2 = aux_next[2u * id + 1u];                               // the dielectric's ray: nothing left to draw
                    launch_ray(tr, make_ray(mk3(o2.x, o2.y, o2.z), mk3(next0.x, next0.y, next0.z)), bounce, true);
                } else if ((int)bounce <= FP().bounce_limit) {                               // :88 loop condition
                    dir_prim = checked(prim_of(id, W), SC().n_tris, 0x1000u);
                    const f3 fn = fetch_face_normal((int)dir_prim);
                    dir_back = (W & kBackFace) != 0;
                    const f3 n = dir_back ? (-1.f * fn) : fn;
                    need_dir = true; dir_origin = mk3(u2f(A.x), u2f(A.y), u2f(A.z)); dir_normal = n;
                    dir_seed = E.w; dir_bounce = bounce; dir_tries = 0;
                    if (EXT && next0.w == 1.0f) { dir_metal = true; dir_normal = mk3(next0.x, next0.y, next0.z); dir_rough = aux_next[2u * id + 1u].x; }
                } else {
                    st1(qA + id * 16u + 12u, 0u);        // the path ends without reaching the sky
                    st1(meta_at(id), make_meta(kNoPrim, 0u, kHasSample));
                    dest = QE;
                }
            }
            draw_and_launch(false);
        } else {
            // ============ E: finish the path, store its sample; deal a new sample, primary ray (RayGen.cuh:63-108,165-171) ============
            if (active) {
                if (ld1(meta_at(id)) & kHasSample) {
                    const uint4 E = aux[id];
                    const uint32_t slot = aux_slot[id];
                    const uint4 B = ld4(qB + id * 16u);
                    const float hit_t = u2f(ld1(qA + id * 16u + 12u));
                    f3 light = mk3(0, 0, 0);
                    if (SUN || (EXT && FP().ext_emissive)) { const float4 l = aux_light[id]; light = mk3(l.x, l.y, l.z); }
                    if (!(hit_t < FLT_MAX))                                                // miss: :99-108
                        light = light + sky_model(mk3(u2f(B.x), u2f(B.y), u2f(B.z)), ld3(FP().sky_color)) * mk3(u2f(E.x), u2f(E.y), u2f(E.z)) * FP().sky_intensity;
                    if (FP().tone_mapping) light = uncharted2_filmic(light, FP().exposure);    // :165-169 (wave-uniform branches)
                    if (FP().gamma_correction) light = gamma_correction(light);
                    const uint32_t slot_ok = held(slot, FP().width * FP().local_rows * FP().n_frames, 0x8000u);
                    if (FP().inline_resolve) accumulate_and_resolve(FP(), slot_ok, light);     // one frame in the launch: slot = pixel
                    else ka->samples[slot_ok] = make_float4(light.x, light.y, light.z, 0.0f);
                }
            }
            // Sample ids.  The first P of a workgroup are fixed (workgroup w starts with [w*P, (w+1)*P): the slots that wait in E's
            // ring when the kernel starts, positions < P) -- no 8 000 waves hammering one counter while the launch ramps up.  The
            // ids above gridDim.x * P come from kSampleShards counters, each on a cache line of its own: counter k hands out
            // j = 0, 1, 2, ... and stands for the ids ((j / 64) * kSampleShards + k) * 64 + j % 64 (chunks of 64 dealt round robin
            // to the counters), one global atomic per batch.  A lone launch on the whole chip makes ~80 M such atomics a second;
            // on ONE address that rate, not the arithmetic, set its time (three launches in flight, each with its own counter,
            // ran 22 % faster per frame than one alone).  A wave starts on the counter of its workgroup and moves on when that
            // one has run dry; a slot retires only after every counter has been found dry.
            const bool initial = active && head + (uint32_t)lane < P;
            const uint32_t n_static = gridDim.x * P;
            const uint32_t n_dynamic = PA().total_samples > n_static ? PA().total_samples - n_static : 0u;
            uint32_t sid = initial ? blockIdx.x * P + head + (uint32_t)lane : 0xFFFFFFFFu;
            bool need = active && !initial;
            for (uint32_t attempt = 0; attempt < kSampleShards; ++attempt) {
                const unsigned long long m = pp_ballot(need);
                if (m == 0) break;
                unsigned int base = 0;
                if (lane == 0) base = atomicAdd(ka->sample_counter + my_shard * kSampleShardStride, (unsigned int)__popcll(m));
                base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                if (need) {
                    const uint32_t j = base + (uint32_t)pp_rank(m);
                    const uint32_t d = ((((j >> 6) * kSampleShards) + my_shard) << 6) | (j & 63u);
                    if (d < n_dynamic) { sid = n_static + d; need = false; }
                }
                if (pp_ballot(need) != 0) my_shard = (my_shard + 1u) & (kSampleShards - 1u);        // this counter has run dry
            }
            if (active) {
                if (sid >= PA().total_samples) {
                    st1_shared(ctrl + kCtrlExhausted, 1u);
                    lds_add(ctrl + kCtrlLive, 0xFFFFFFFFu);                                // this pool slot retires
                } else {
                    // sample id -> (tile, frame, pixel): chunk = 64 samples = one 8x8 tile of one frame, frame-major, tile rows
                    // visited with a stride (FP().row_step) -- the work order of wave_queue (DRT_CHUNK_ORDER 4)
                    const uint32_t my_chunk = sid >> 6, my_k = sid & 63u;
                    const uint32_t n_tiles_all = PA().n_chunks / FP().n_frames;
                    const uint32_t f_rel = my_chunk / n_tiles_all, tile = my_chunk - f_rel * n_tiles_all;
                    uint32_t ty = tile / PA().tiles_x;
                    const uint32_t tx = tile - ty * PA().tiles_x;
                    ty = (ty * FP().row_step) % (n_tiles_all / PA().tiles_x);
                    const uint32_t x = tx * 8u + (my_k & 7u), ly = ty * 8u + (my_k >> 3);
                    if (x < FP().width && ly < FP().local_rows) {
                        const uint32_t y = ((ly / FP().stripe_rows) * FP().world + FP().rank) * FP().stripe_rows + (ly % FP().stripe_rows);
                        const uint32_t slot = f_rel * (FP().width * FP().local_rows) + ly * FP().width + x;
                        f2 screen_uv;
                        screen_uv.x = ((float)x / (float)FP().width) * 2 - 1;
                        screen_uv.y = ((float)y / (float)FP().height) * 2 - 1;
                        uint32_t seed = x + y * FP().width;
                        seed *= FP().frame_first + f_rel;
                        count(C_SAMPLES);
                        const Ray ray = camera_get_ray(FP(), screen_uv, seed);
                        launch_ray(tr, ray, 0u, FP().bounce_limit >= 0);
                        aux[id] = make_uint4(f2u(1.0f), f2u(1.0f), f2u(1.0f), seed);
                        aux_slot[id] = slot;
                        if (SUN || (EXT && FP().ext_emissive)) aux_light[id] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    } else {
                        st1(meta_at(id), HBM ? 0u : kNoPrim);       // a sample id outside the image (partial tile): the slot asks again
                        dest = QE;
                    }
                }
            }
        }
        if (q < QT0 || q >= QB) {
            // ============ the N loop: the lanes of an N batch, and the rays this batch has just launched ============
            // One visit per iteration (pop_step); a lane leaves when it stands on a leaf or its stack is empty.  Lanes that are
            // done wait for the others only while enough of them are still popping; the rest goes (back) to the N queue.
            for (uint32_t it = 0;; ++it) {
                const bool go = tr.on && dest < 0 && (tr.have_top || tr.sp > 0);
                const unsigned long long m_go = pp_ballot(go);
                // (a launching batch: the root's visit always; further visits while enough of its lanes need one -- the others
                // have their leaf or their miss already and wait)
                if (m_go == 0 || it >= (q == QN ? k_n_loop : k_n_fuse_loop) || (it > 0 && (uint32_t)__popcll(m_go) < (q == QN ? k_n_min_lanes : k_n_fuse_min))) break;
                if (STATS) { s_work[0]++; s_work[1] += (unsigned long long)__popcll(m_go); }
                if (go) dest = pop_step(tr.ray, tr.hit_t, tr.sp, id, tr.cur, tr.end, tr.top, tr.have_top, SUN && tr.shadow);
            }
            if (tr.on || tr.fresh) {
                if (dest < 0) { spill_top(tr.sp, id, tr.top, tr.have_top); tr.cur = tr.end = 0; dest = tr.sp > 0 ? QN : after_traversal(tr.hit_t, tr.shadow); }
                if (tr.fresh) {
                    st4(qA + id * 16u, make_uint4(f2u(tr.ray.orig.x), f2u(tr.ray.orig.y), f2u(tr.ray.orig.z), tr.hit_store));
                    if (HBM) {
                        st4(qB + id * 16u, make_uint4(f2u(tr.ray.dir.x), f2u(tr.ray.dir.y), f2u(tr.ray.dir.z), tr.cur));
                        st1(meta_at(id), tr.meta | ((uint32_t)tr.sp << 16));
                        st1(meta_at(id) + 8u, tr.end);
                    } else {
                        st4(qB + id * 16u, make_uint4(f2u(tr.ray.dir.x), f2u(tr.ray.dir.y), f2u(tr.ray.dir.z), tr.cur | (tr.end << 12) | ((uint32_t)tr.sp << 24)));
                        st1(meta_at(id), tr.meta);
                    }
                } else store_trav(id, tr.sp, tr.cur, tr.end, tr.meta);
            }
        }
        if (!HBM) __builtin_amdgcn_s_setprio(1);
        lds_release();
        push_group(dest, id);
        if ((HBM || STATS) && pp_ballot(violations != 0) != 0) { if (violations != 0 && PA().status) atomicOr(PA().status, violations); violations = 0; }
        if (STATS && lane == 0) lds_add64(ctrl + kCtrlStats + (uint32_t)q * 24u + 16u, __builtin_amdgcn_s_memtime() - s_t1);
    }

    if (FP().span && lane == 0) atomicMax(&FP().span[1], (unsigned long long)wall_clock64());
    if (STATS && FP().counters) {
        // (wave sums by DPP-free butterfly through ds_bpermute would save atomics; a counting launch is not timed)
        for (int k = 0; k < C_COUNT; k++) if (work[k]) atomicAdd(&FP().counters[k == C_TRIES ? 23 : k], (unsigned long long)work[k]);     // (drt_counters: the ten work counters, 13 words of wave_queue's phase statistics, sampler_tries)
    }
    if (STATS && PA().stats) {                               // (every wave gets here: the loop ends for all of them, by `live == 0` or by the abort flag)
        __syncthreads();
        if (tid < 3 * kNQ) atomicAdd(&PA().stats[tid], ld64(ctrl + kCtrlStats + (uint32_t)tid * 8u));
    }
    if (STATS && PA().stats && lane == 0) {
        atomicAdd(&PA().stats[3 * kNQ], s_claim); atomicAdd(&PA().stats[3 * kNQ + 1], s_idle); atomicAdd(&PA().stats[3 * kNQ + 2], s_lost);
        atomicAdd(&PA().stats[3 * kNQ + 3], __builtin_amdgcn_s_memtime() - s_t_start);
        for (int k = 0; k < 6; k++) atomicAdd(&PA().stats[3 * kNQ + 7 + k], s_work[k]);
        atomicAdd(&PA().stats[3 * kNQ + 4], s_fail); atomicAdd(&PA().stats[3 * kNQ + 5], s_fail_ticks); atomicAdd(&PA().stats[3 * kNQ + 6], s_idle_ticks);
    }
}

}  // namespace

#ifdef DRT_POOL_EXT_TU
// ---- this translation unit (kernel_path_pool_ext.o): the material-model variants of the kernel, nothing else ----
const void *path_pool_ext_kernel(int flags) {        // (untyped: the argument struct lives in each translation unit's anonymous namespace)
    if (flags & 32) switch (flags & 14) {
    case 0: return reinterpret_cast<const void *>(path_pool_kernel<48>);
    case 2: return reinterpret_cast<const void *>(path_pool_kernel<50>);
    case 4: return reinterpret_cast<const void *>(path_pool_kernel<52>);
    case 6: return reinterpret_cast<const void *>(path_pool_kernel<54>);
    case 8: return reinterpret_cast<const void *>(path_pool_kernel<56>);
    case 10: return reinterpret_cast<const void *>(path_pool_kernel<58>);
    case 12: return reinterpret_cast<const void *>(path_pool_kernel<60>);
    default: return reinterpret_cast<const void *>(path_pool_kernel<62>);
    }
    switch (flags & 14) {
    case 0: return reinterpret_cast<const void *>(path_pool_kernel<16>);
    case 2: return reinterpret_cast<const void *>(path_pool_kernel<18>);
    case 4: return reinterpret_cast<const void *>(path_pool_kernel<20>);
    case 6: return reinterpret_cast<const void *>(path_pool_kernel<22>);
    case 8: return reinterpret_cast<const void *>(path_pool_kernel<24>);
    case 10: return reinterpret_cast<const void *>(path_pool_kernel<26>);
    case 12: return reinterpret_cast<const void *>(path_pool_kernel<28>);
    default: return reinterpret_cast<const void *>(path_pool_kernel<30>);
    }
}
}  // namespace drt
#else
const void *path_pool_ext_kernel(int flags);                    // kernel_path_pool_ext.o

// Leaf step counts (two triangles per step) -> upper bounds of the first three T queues, chosen so that the steps a
// batch wastes on shorter leaves (every lane runs as long as the batch's longest leaf) are fewest, each leaf weighted by
// its size.  At most a dozen distinct values: exhaustive.
void path_pool_leaf_classes(const std::vector<LeafRange> &leaves, uint32_t out[3]) {
    std::vector<uint32_t> steps;
    for (const LeafRange &l : leaves) steps.push_back(((uint32_t)l.count + 1u) / 2u);
    std::vector<uint32_t> distinct = steps;
    std::sort(distinct.begin(), distinct.end());
    distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
    out[0] = out[1] = out[2] = distinct.empty() ? 0u : distinct.back();
    if (distinct.size() <= 1) return;
    const size_t m = distinct.size();
    auto waste = [&](uint32_t b0, uint32_t b1, uint32_t b2) {
        unsigned long long w = 0;
        for (uint32_t s : steps) {
            const uint32_t top = s <= b0 ? b0 : (s <= b1 ? b1 : (s <= b2 ? b2 : distinct.back()));
            w += (unsigned long long)(top - s) * s;
        }
        return w;
    };
    unsigned long long best = ~0ull;
    for (size_t a = 0; a < m; a++)
        for (size_t b = a; b < m; b++)
            for (size_t c = b; c < m; c++) {
                const unsigned long long w = waste(distinct[a], distinct[b], distinct[c]);
                if (w < best) { best = w; out[0] = distinct[a]; out[1] = distinct[b]; out[2] = distinct[c]; }
            }
}

// Does the pool apply, and in which build: *hbm_scene = the traversal data does not fit LDS next to a pool and is read from global memory.
bool path_pool_supports(const SceneView &sc, const FrameParams &fp, int bvh_depth, size_t scene_lds_bytes, bool *hbm_scene) {
    if (fp.render_mode != 0) return false;                                                // no debug views (wave_queue's general build)
    if (fp.bounce_limit > 60000 || bvh_depth > 200) return false;                         // bounce index in 16 bits, stack height in 8
    if ((fp.ext_emissive || fp.ext_specular || fp.ext_transmission) && fp.bounce_limit > 30000) return false;      // (material-model builds: 15 bits)
    if (sc.root_ref == kNoNode) return false;
    static const size_t scene_limit = std::getenv("DRT_POOL_SCENE_KB") ? (size_t)std::atoi(std::getenv("DRT_POOL_SCENE_KB")) * 1024 : kLdsSceneBytes;
    static const bool hbm_allowed = !(std::getenv("DRT_POOL_HBM") && std::atoi(std::getenv("DRT_POOL_HBM")) == 0);
    const uint32_t stack_entries = (uint32_t)std::max(bvh_depth - 1, 1);
    // lds-scene build: 12-bit triangle indices, and a pool of at least 256 paths has to fit the CU's LDS next to the scene copy
    // (a very deep tree's stacks may not leave room)
    bool in_lds = scene_lds_bytes <= scene_limit && sc.n_tris < 4095u;
    if (in_lds && pool_layout(512u, 512u, stack_entries, pool_scene_bytes(sc), 0u, (uint32_t)kNQ).total > 160u * 1024u &&
        (scene_lds_bytes > kLdsSceneBytes || pool_layout(256u, 256u, stack_entries, pool_scene_bytes(sc), 0u, (uint32_t)kNQ).total > 160u * 1024u)) in_lds = false;
    if (hbm_scene) *hbm_scene = !in_lds;
    if (in_lds) return true;
    return hbm_allowed && sc.n_inner < 32768u && sc.n_leaves < 32768u &&               // 16-bit node references on the stacks
           pool_layout(256u, 256u, stack_entries, 0u, 0u, (uint32_t)kNQ, 16u, 6u).total <= 160u * 1024u;
}

hipError_t launch_path_pool(const SceneView &sc, const FrameParams &fp, int bvh_depth, bool scene_has_alpha, bool hbm_scene, const uint32_t t_class[3], const PoolTuning &tune,
                            PoolScratch &scratch, unsigned int *sample_counter, void *samples, unsigned int *status, int num_cus,
                            hipStream_t stream, const char **kernel_name, int *launch_shape) {
    if (fp.width == 0 || fp.local_rows == 0 || fp.n_frames == 0) return hipSuccess;
    const int env_threads = tune.threads, env_paths = tune.paths, env_fill = tune.min_fill, env_patience = tune.patience;
    const uint32_t tiles_x = (fp.width + 7) / 8, tiles_y = (fp.local_rows + 7) / 8;
    const uint64_t n_chunks = (uint64_t)tiles_x * tiles_y * fp.n_frames;
    if (n_chunks * 64ull > 0xF0000000ull) return hipErrorInvalidValue;       // (room for the ids the sharded counters hand out past the end)
    // Stack slots: the far siblings of the nodes on the way down, one per level below the root -- the near child of a visit stays
    // in registers and is spilled only where it is an interior node, so never more than levels - 1 entries.  (Smaller entries were
    // tried on room, where 64 more paths in the pool are worth 3-5 %: 6 bytes {distance, parent << 1 | which child} cost 3.5 % at
    // equal pool size and end level at 1408 paths against 1216; 2 bytes with the distance computed again at the pop cost 7 %.)
    // (And once more after the hbm-scene build had them: 6-byte entries {distance, 16-bit reference} with the leaf ranges back in an
    // LDS table -- room 1408 paths, 44.0 ms against 43.6 with 8-byte entries and 1216 paths; cornell's pools come out as 3 x 704, 3.63 ms
    // against 3.47.  The lds-scene build keeps its 8-byte entries.)
    // (Rings of exactly P entries instead of the next power of two -- positions counted modulo a multiple of P, slot = position
    // mod P by multiplication -- were tried too: room's pool grows from 1216 to 1344 paths, and the extra arithmetic in every claim
    // and push costs the 3 % that buys.)
    const uint32_t stack_entries = (uint32_t)std::max(bvh_depth - 1, 1);
    // The levels of the stacks kept in LDS (tune.stack_lds; 0 = chosen below, with the pool's shape)
    uint32_t stack_lds = tune.stack_lds > 0 ? std::min<uint32_t>((uint32_t)tune.stack_lds, stack_entries) : stack_entries;
    const uint32_t scene_bytes = hbm_scene ? 0u : pool_scene_bytes(sc);
    const uint32_t word_bytes = hbm_scene ? 16u : 4u, stack_entry_bytes = hbm_scene ? 6u : 8u;
    // the shading records go to LDS too when they are small (cornell: 1.2 KB): B's load chain triangle -> material ->
    // texture header then runs through LDS instead of three dependent HBM / L2 round trips
    uint32_t cold_bytes = pool_cold_bytes(sc) <= (tune.cold_lds_kb >= 0 ? (uint32_t)tune.cold_lds_kb * 1024u : 4096u) ? pool_cold_bytes(sc) : 0u;
    // Workgroups per CU, paths per pool, threads, stack levels in LDS.  Two things keep a CU busy: resident waves (at most 24 here:
    // 6 per SIMD at this kernel's 106 SGPRs; 16 in one workgroup) and ~1.3 paths per lane, so that a wave finds a full batch
    // when it looks for one (768 -> 1024 paths at 768 threads: -13 % time; beyond ~1.35 per lane more paths buy nothing).  What
    // stands in the way is LDS: rings are sized to the next power of two (1024 paths per workgroup is a sweet spot), and every
    // stack level costs 8 (6) bytes per path.  A deep tree therefore keeps only the BOTTOM k levels of its stacks in LDS and the
    // rest in HBM (path_pool_kernel<FLAGS | 32>): a path is deep in the tree on few of its visits, and what the freed LDS holds
    // is worth more -- paths (dense_monkey, 14 levels: all in LDS 1024 paths, 9 513 Msamples/s; 8 levels 1344 paths, 10 593;
    // cs16_dust 1 094 -> 1 231) or a second workgroup (room 4K, 7 levels, scene copy in LDS: 1 x 1216 paths x 16 waves 847 Msamples/s;
    // 6 levels, 1 x 1344 x 16 waves 860; ONE level, 2 x 1024 x 12 waves 1 000).  Every shape is scored by the lanes it can keep
    // busy, min(resident lanes, paths / 1.19); ties go to more levels in LDS, then to fewer workgroups.  (1.19, not 1.3: a pool of
    // 1216 paths per 1024 threads gains nothing from stack levels in HBM -- suzanne with sunlight, whose shadow traversals push more
    // entries: 2.56 ms with all 9 levels in LDS, 2.66 with 8 and 1280 paths.)
    // (tools/experiments/r03/sweep_stack_lds.sh, room_two_pools.sh)
    const uint32_t n_rings = fp.enable_sunlight ? (uint32_t)kNQ : (uint32_t)kNQ - 1u;
    const uint32_t stats_bytes = (tune.stats || fp.counters) ? kStatsBytes : 0u;         // (the statistics build keeps its per-queue sums in LDS)
    auto lds_for = [&](uint32_t paths, uint32_t &cap, uint32_t levels) {
        cap = 64;
        while (cap < paths) cap *= 2;
        return pool_layout(paths, cap, levels, scene_bytes, cold_bytes, n_rings, word_bytes, stack_entry_bytes, stats_bytes).total;
    };
    const int flags = ((tune.stats || fp.counters) ? 1 : 0) | (fp.enable_sunlight ? 2 : 0) | (scene_has_alpha ? 4 : 0) | (hbm_scene ? 8 : 0);
    const int max_threads = pool_max_threads(flags);
    auto threads_for = [&](uint32_t paths, int g) {
        // hbm-scene: every step waits for L2, so all the waves a workgroup can have (even a few more lanes than paths: measured on cs16_dust)
        if (hbm_scene) return std::min(max_threads, 1536 / g / 64 * 64);
        return std::max(256, std::min<int>({ max_threads, (int)paths / 64 * 64, 1536 / g / 64 * 64 }));
    };
    uint32_t P = 0, ring_cap = 64;
    int groups = 1;
    if (env_paths > 0) {
        P = std::max<uint32_t>(64u, std::min<uint32_t>((uint32_t)env_paths / 64u * 64u, kMaxPoolPaths));
        while (P > 64u && lds_for(P, ring_cap, stack_lds) > 160u * 1024u) P -= 64u;
    } else {
        // (hbm-scene: two pools of 12 waves per CU with the narrow kernels, one of 16 with the wide statistics builds)
        auto best_shape = [&]() {
            uint32_t best_score = 0, best_levels = stack_lds;
            const uint32_t k_first = stack_lds, k_last = tune.stack_lds > 0 ? stack_lds : 1u;
            for (uint32_t k = k_first; k >= k_last; --k)
                for (int g = 1; g <= (hbm_scene ? (pool_narrow(flags) ? 2 : 1) : 3); g++)
                    for (uint32_t paths = kMaxPoolPaths; paths >= 256u; paths -= 64u) {
                        uint32_t cap;
                        if (lds_for(paths, cap, k) * (uint32_t)g > 160u * 1024u) continue;
                        const uint32_t lanes = std::min<uint32_t>(1536u, (uint32_t)g * (uint32_t)threads_for(paths, g));
                        const uint32_t score = std::min<uint32_t>(lanes, paths * (uint32_t)g * 16u / 19u);
                        if (score > best_score) { best_score = score; P = paths; groups = g; best_levels = k; }
                        break;
                    }
            stack_lds = best_levels;
        };
        best_shape();
        // path_pool_supports sizes its smallest pool without the staged shading records: a small scene under a very deep tree may
        // leave no room for them -- the records then stay in global memory rather than the launch failing
        if (P == 0 && cold_bytes > 0) { cold_bytes = 0; best_shape(); }
        if (P == 0) return hipErrorInvalidValue;
    }
    (void)lds_for(P, ring_cap, stack_lds);
    const PoolLayout lay = pool_layout(P, ring_cap, stack_lds, scene_bytes, cold_bytes, n_rings, word_bytes, stack_entry_bytes, stats_bytes);
    if (lay.total > 160u * 1024u) return hipErrorInvalidValue;
    const int threads = env_threads > 0 ? std::max(64, std::min(env_threads, max_threads) / 64 * 64) : threads_for(P, groups);
    typedef void (*PoolKernel)(const PoolArgs);
    static const PoolKernel kernels[32] = { path_pool_kernel<0>, path_pool_kernel<1>, path_pool_kernel<2>, path_pool_kernel<3>,
                                            path_pool_kernel<4>, path_pool_kernel<5>, path_pool_kernel<6>, path_pool_kernel<7>,
                                            path_pool_kernel<8>, path_pool_kernel<9>, path_pool_kernel<10>, path_pool_kernel<11>,
                                            path_pool_kernel<12>, path_pool_kernel<13>, path_pool_kernel<14>, path_pool_kernel<15>,
                                            // (deep trees: the stacks' upper levels in HBM)
                                            path_pool_kernel<32>, path_pool_kernel<33>, path_pool_kernel<34>, path_pool_kernel<35>,
                                            path_pool_kernel<36>, path_pool_kernel<37>, path_pool_kernel<38>, path_pool_kernel<39>,
                                            path_pool_kernel<40>, path_pool_kernel<41>, path_pool_kernel<42>, path_pool_kernel<43>,
                                            path_pool_kernel<44>, path_pool_kernel<45>, path_pool_kernel<46>, path_pool_kernel<47> };
    static const char *const names[16] = { "path_pool<lean,lds-scene>", "path_pool<lean+sun,lds-scene>", "path_pool<lean+alpha,lds-scene>",
                                           "path_pool<lean+alpha+sun,lds-scene>", "path_pool<lean,hbm-scene>", "path_pool<lean+sun,hbm-scene>",
                                           "path_pool<lean+alpha,hbm-scene>", "path_pool<lean+alpha+sun,hbm-scene>",
                                           "path_pool<materials,lds-scene>", "path_pool<materials+sun,lds-scene>", "path_pool<materials+alpha,lds-scene>",
                                           "path_pool<materials+alpha+sun,lds-scene>", "path_pool<materials,hbm-scene>", "path_pool<materials+sun,hbm-scene>",
                                           "path_pool<materials+alpha,hbm-scene>", "path_pool<materials+alpha+sun,hbm-scene>" };
    const bool material_model = fp.ext_emissive || fp.ext_specular || fp.ext_transmission;
    // (the material-model variants live in a translation unit of their own, kernel_path_pool_ext.o: no statistics builds of them)
    const bool deep = stack_lds < stack_entries;
    const void *const kernel = material_model ? path_pool_ext_kernel((flags & ~1) | (deep ? 32 : 0)) : reinterpret_cast<const void *>(kernels[flags + (deep ? 16 : 0)]);
    if (lay.total > 64u * 1024u) {
        hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lay.total);
        if (e != hipSuccess) return e;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lay.total) != hipSuccess || per_cu < 1) per_cu = 1;
    PoolParams pp;
    pp.P = P; pp.ring_cap = ring_cap; pp.stack_entries = stack_entries; pp.stack_lds = stack_lds;
    pp.ring_shift = 0;
    while ((1u << pp.ring_shift) < ring_cap) pp.ring_shift++;
    if (P > kMaxPoolPaths || (1u << pp.ring_shift) != ring_cap || ring_cap < P) return hipErrorInvalidValue;   // 12-bit ids, power-of-two rings
    pp.total_samples = (uint32_t)(n_chunks * 64ull); pp.n_chunks = (uint32_t)n_chunks; pp.tiles_x = tiles_x;
    pp.t_class[0] = t_class[0]; pp.t_class[1] = t_class[1]; pp.t_class[2] = t_class[2];
    pp.min_fill = (uint32_t)std::max(1, std::min(env_fill, 64)); pp.patience = (uint32_t)std::max(0, env_patience);
    pp.n_loop = (uint32_t)std::max(1, tune.n_loop); pp.n_min_lanes = (uint32_t)std::max(1, std::min(tune.n_min_lanes, 64));
    pp.n_fuse_loop = (uint32_t)std::max(1, tune.n_fuse_loop); pp.n_fuse_min = (uint32_t)std::max(1, std::min(tune.n_fuse_min, 64));
    pp.cold_in_lds = cold_bytes > 0 ? 1u : 0u;
    pp.dir_tries = (uint32_t)std::max(1, tune.dir_tries);
    pp.status = status; pp.stats = tune.stats;
    // one pool fills with P samples at once: never more workgroups than that leaves work for.  When the caller keeps several
    // launches in flight (drt_renderer_set_frames_in_flight) each gets its share of the workgroup slots, so that they run side
    // by side and the ramp-up and drain of one overlap the steady state of the others instead of queueing behind a full grid.
    uint64_t want = std::min<uint64_t>((uint64_t)num_cus * per_cu, std::max<uint64_t>(1, (n_chunks * 64ull + P - 1) / P));
    // (only launches short enough for ramp-up and drain to matter -- dense_monkey 16 spp, 145 refills: 7.3 -> 8.8 Gsamples/s; a launch
    // that refills its pools two hundred times runs best on the whole chip: room 4K / 64 spp with three frames in flight, 1029 ms
    // per step on a third of the grid each against 694 with full grids)
    // (DRT_POOL_SHARE_GRID = the number of pool refills below which a launch shares; 1 = the default of 160, 0 = never)
    const uint64_t share_refills = tune.share_grid == 1 ? 160ull : (uint64_t)std::max(0, tune.share_grid);
    if (fp.frames_in_flight > 1 && share_refills > 0 && n_chunks * 64ull < share_refills * (uint64_t)num_cus * per_cu * P)
        want = std::min<uint64_t>(want, std::max<uint64_t>(1, ((uint64_t)num_cus * per_cu + fp.frames_in_flight - 1) / (uint64_t)fp.frames_in_flight));
    // the part of the path state that lives in HBM: 36 bytes per pool slot of every workgroup (16 of them used by sunlight only)
    const size_t slots = (size_t)num_cus * per_cu * P;
    if (slots > scratch.slots) {
        if (scratch.aux) (void)hipFree(scratch.aux);
        if (scratch.aux_slot) (void)hipFree(scratch.aux_slot);
        if (scratch.aux_light) (void)hipFree(scratch.aux_light);
        scratch.aux = nullptr; scratch.aux_slot = nullptr; scratch.aux_light = nullptr; scratch.slots = 0;
        hipError_t ea = hipMalloc(&scratch.aux, slots * 16);
        if (ea != hipSuccess) return ea;
        ea = hipMalloc(&scratch.aux_slot, slots * 4);
        if (ea != hipSuccess) return ea;
        ea = hipMalloc(&scratch.aux_light, slots * 16);
        if (ea != hipSuccess) return ea;
        scratch.slots = slots;
    }
    if (material_model && fp.enable_sunlight && slots > scratch.next_slots) {            // (only the material model with sunlight uses it)
        if (scratch.aux_next) (void)hipFree(scratch.aux_next);
        scratch.aux_next = nullptr; scratch.next_slots = 0;
        const hipError_t ea = hipMalloc(&scratch.aux_next, slots * 32);
        if (ea != hipSuccess) return ea;
        scratch.next_slots = slots;
    }
    if (stack_lds < stack_entries && slots * (stack_entries - stack_lds) > scratch.stack_slots) {
        if (scratch.aux_stack) (void)hipFree(scratch.aux_stack);
        scratch.aux_stack = nullptr; scratch.stack_slots = 0;
        const hipError_t ea = hipMalloc(&scratch.aux_stack, slots * (stack_entries - stack_lds) * 8);
        if (ea != hipSuccess) return ea;
        scratch.stack_slots = slots * (stack_entries - stack_lds);
    }
    pp.aux_stack = static_cast<uint2 *>(scratch.aux_stack);
    pp.aux = static_cast<uint4 *>(scratch.aux); pp.aux_slot = static_cast<uint32_t *>(scratch.aux_slot);
    pp.aux_light = static_cast<float4 *>(scratch.aux_light);
    pp.aux_next = static_cast<float4 *>(scratch.aux_next);
    hipError_t e = hipSuccess;                 // (sample_counter: kPoolSampleShards zeroed counters, kPoolSampleShardStride words apart -- drt_capi.cpp hands out zeroed blocks)
    if (kernel_name) *kernel_name = names[(flags >> 1) + (material_model ? 8 : 0)];
    if (launch_shape) { launch_shape[0] = (int)(stack_entries | (stack_lds << 8)); launch_shape[1] = per_cu; launch_shape[2] = (int)(lay.total / 1024); launch_shape[3] = threads; launch_shape[4] = (int)P; }
    FrameParams fq = fp;
    fq.inline_resolve = fp.n_frames == 1 ? 1 : 0;
    PoolArgs args;
    args.sc = sc; args.fp = fq; args.pp = pp; args.sample_counter = sample_counter; args.samples = static_cast<float4 *>(samples);
    void *kernel_args[] = { &args };
    e = hipLaunchKernel(kernel, dim3((unsigned)want), dim3(threads), kernel_args, lay.total, stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess || fq.inline_resolve) return e;
    return launch_resolve(fp, samples, stream);
}

}  // namespace drt
#endif  // DRT_POOL_EXT_TU
