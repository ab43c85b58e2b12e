// kernel_wave_queue.hip -- "wave_queue": persistent wave64 path tracer with lane refill and phase voting,
// plus the ordered resolve kernel.
//
// Same arithmetic as pixel_walk (render_kernels.hip) and the same per-lane order of node visits and
// triangle tests as the reference (BVH/BVHTraversal.cuh:14-134), so images are bit-identical; what changes
// is how the 64 lanes of a wave are kept busy:
//
//   * PERSISTENT WAVES + SAMPLE QUEUE.  The grid is sized to the chip, not to the image.  The unit of work is
//     one SAMPLE (pixel x frame index = one RayGen call, RayGen.cuh:63).  A wave pulls chunks of 64 samples
//     (one 8x8 pixel tile, one frame index) from a global atomic counter (one atomic per 64 samples) and deals
//     them to its lanes one by one: a lane that finishes its path takes the next sample at once, so no lane
//     idles because a neighbour's path is longer (paths are 1..depth+1 rays long, RayGen.cuh:88-162), and the
//     end-of-launch tail is one sample per lane, not one pixel x all frames.
//   * ORDERED RESOLVE.  Each sample's colour goes to a float4 slot [frame][pixel] in HBM; resolve_kernel then
//     adds the frames of a pixel in frame order, ((a + c_f) + c_f+1) + ..., exactly the order of
//     accumulation_buffer += fcolor over successive launches (RenderKernel.cu:29-30), and writes accum + RGBA.
//   * PHASE VOTING.  A lane is in one of three traversal/shading states: T = has a leaf triangle to test, N = has a
//     node on its stack, S = needs shading / a new ray / a new sample.  The wave loop ballots the states and runs
//     ONE phase for all lanes in that state; T (one Moller-Trumbore test per lane, the dominant work) keeps running
//     until enough lanes wait for another phase.
//   * SPECULATIVE BOUNCE DIRECTIONS (phase R).  randomUnitSphereVec3 is a rejection loop (~2.9 tries, Random.cu:50-58)
//     whose draws depend only on the seed, not on what the ray hits.  As soon as a ray is launched, the lane knows
//     the seed its bounce direction will be drawn from (RayGen.cuh:91 seed += i), so the tries run in the BACKGROUND,
//     one candidate per R step, for every lane with a direction pending -- whatever its traversal state -- and a
//     hit usually finds its direction ready.  If the ray misses, the direction is simply dropped (the reference never
//     draws it); the RNG stream seen by the path is unchanged.
//   * LDS.  Traversal stack (node reference + entry distance, 8 B) per lane in LDS, entry [level][tid]:
//     conflict-free for ds_read/write_b64.  For scenes whose traversal data (child-box-pair records, leaf
//     ranges, TriHot records) fits kLdsSceneBytes, every workgroup stages it in LDS once and all node /
//     triangle fetches are 32-bit-addressed LDS gathers.
// No MFMA (branchy scalar fp32 / u32).  Citations are relative to /root/reference/DustRayTracer/src/.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <functional>
#include <cstdlib>
#include <vector>

#include "device_access.hpp"
#include "device_math.hpp"
#include "device_scene.hpp"
#include "render_kernels.hpp"

namespace drt {

namespace {

constexpr int kThreads = 256;         // workgroup size; kBigThreads where a bigger group shares its LDS scene copy among more waves
constexpr int kBigThreads = 512;
#ifndef DRT_PRIO_S
#define DRT_PRIO_S 1
#endif
#ifndef DRT_TRIS_PER_STEP
#define DRT_TRIS_PER_STEP 2
#endif
constexpr int kTrianglesPerStep = DRT_TRIS_PER_STEP;      // lean kernel only

// per-lane path stage: what the S phase has to do next for this lane (kNeedDir lanes are served by R)
enum : int { kNeedSample = 0, kTraceDone = 2, kShadowDone = 3, kFinished = 4, kPathDone = 5, kNeedDir = 6 };

struct StackEntry { uint32_t ref; float dist; };

DRT_DEV unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// LDS reads through 32-bit address-space-3 pointers (byte offset within the workgroup's LDS allocation): keeps the
// address arithmetic in 32 bits (v_mad_u32_u24) instead of a 64-bit flat pointer computation.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
#if defined(__HIP_DEVICE_COMPILE__)
DRT_DEV uint4 lds_load4(uint32_t byte_off) {
    const u32x4_t v = *(__attribute__((address_space(3))) const u32x4_t *)byte_off;
    return make_uint4(v.x, v.y, v.z, v.w);
}
DRT_DEV uint2 lds_load2(uint32_t byte_off) {
    const u32x2_t v = *(__attribute__((address_space(3))) const u32x2_t *)byte_off;
    return make_uint2(v.x, v.y);
}
DRT_DEV uint32_t lds_load1(uint32_t byte_off) { return *(__attribute__((address_space(3))) const uint32_t *)byte_off; }
#else       // host pass: never called, only parsed
inline uint4 lds_load4(uint32_t) { return make_uint4(0, 0, 0, 0); }
inline uint2 lds_load2(uint32_t) { return make_uint2(0, 0); }
inline uint32_t lds_load1(uint32_t) { return 0; }
#endif
DRT_DEV int lane_rank(unsigned long long mask) {        // set bits below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// MODE 0: lean (NORMALMODE, no sunlight, no RGBA texture, no counters); 1: every setting honoured
// at run time; 2: = 1 + exact work counters; 3: = lean + the alpha test of AnyHit.cuh on closest-hit candidates (scenes
// with RGBA textures: the reference's cut-out foliage / fences); 4: = lean + sunlight (a shadow ray per hit); 5: = 3 + 4
// stack_entries = traversal stack slots per lane (the BVH's depth): the LDS a workgroup takes is exactly what its tree needs
#ifdef DRT_WAVES_PER_EU      // experiments only: force the register budget of that many waves per SIMD
#define DRT_OCCUPANCY_ATTR __attribute__((amdgpu_waves_per_eu(DRT_WAVES_PER_EU, DRT_WAVES_PER_EU)))
#else
#define DRT_OCCUPANCY_ATTR
#endif
template <int MODE, bool LDS_SCENE, bool REF16, int TRIS>
__global__ __launch_bounds__(kBigThreads) DRT_OCCUPANCY_ATTR void wave_queue_kernel(const SceneView sc, const FrameParams fp,
                                                              unsigned int *chunk_counter, uint32_t n_chunks, uint32_t tiles_x,
                                                              float4 *samples, uint32_t stack_entries) {
    constexpr bool GENERAL = MODE == 1 || MODE == 2;
    constexpr bool COUNT = MODE == 2;
    constexpr bool ALPHA = MODE == 3 || MODE == 5;
    constexpr bool SUN = MODE == 4 || MODE == 5;          // lean + the sun's shadow ray at every hit (RayGen.cuh:124-128)
    constexpr bool SHADOWS = GENERAL || SUN;              // shadow traversals (RayTest, BVHTraversal.cuh:76-134) can occur
    extern __shared__ uint4 lds_raw[];
    const int tid = threadIdx.x;
    const uint32_t wg = blockDim.x;                                          // 256 or 512 (launch_one)
    // Traversal stack, entry [level][tid]: 8 bytes (node reference + entry distance), or -- when that is what limits the waves
    // per CU and every reference fits 15 bits + the leaf bit -- 4 + 2 bytes in two arrays (REF16, chosen by launch_one)
    // (addresses are formed from the scalar bases at every access: no per-lane pointer lives across the kernel)
    StackEntry *const stack_base = reinterpret_cast<StackEntry *>(lds_raw);
    float *const stack_dist = reinterpret_cast<float *>(lds_raw);
    unsigned short *const stack_ref = reinterpret_cast<unsigned short *>(reinterpret_cast<float *>(lds_raw) + stack_entries * wg);
    auto stack_store = [&](int level, StackEntry e) {
        const uint32_t at = __umul24((uint32_t)level, wg) + (uint32_t)tid;
        if (REF16) { stack_dist[at] = e.dist; stack_ref[at] = (unsigned short)((e.ref & 0x7fffu) | ((e.ref >> 16) & 0x8000u)); }
        else stack_base[at] = e;
    };
    auto stack_load = [&](int level) {
        const uint32_t at = __umul24((uint32_t)level, wg) + (uint32_t)tid;
        StackEntry e;
        if (REF16) { const uint32_t r = stack_ref[at]; e.ref = (r & 0x7fffu) | ((r & 0x8000u) << 16); e.dist = stack_dist[at]; }
        else e = stack_base[at];
        return e;
    };
    const int lane = tid & 63;
    // The kernel's own execution span (first wave in to last wave out, constant-rate clock): what a profiler reports as
    // the kernel's duration, also when launches from several streams share the GPU and stream events include queueing.
    if (fp.span && lane == 0) atomicMax(&fp.span[0], ~(unsigned long long)wall_clock64());

    // ---- scene source: LDS copy (indexed in uint4 units from lds_raw) or HBM ----
    const uint32_t kSceneBase = stack_entries * (wg * (REF16 ? 6u : (uint32_t)sizeof(StackEntry)) / 16u);
    const uint32_t hot_base = kSceneBase + sc.n_inner * 4u;
    const uint32_t leaf_base = hot_base + sc.n_tris * 3u;
    if (LDS_SCENE) {
        const uint4 *g_inner = reinterpret_cast<const uint4 *>(sc.inner);
        const uint4 *g_hot = reinterpret_cast<const uint4 *>(sc.tri_hot);
        for (uint32_t i = tid; i < sc.n_inner * 4u; i += wg) lds_raw[kSceneBase + i] = g_inner[i];
        for (uint32_t i = tid; i < sc.n_tris * 3u; i += wg) lds_raw[hot_base + i] = g_hot[i];
        LeafRange *l_leaves = reinterpret_cast<LeafRange *>(lds_raw + leaf_base);
        for (uint32_t i = tid; i < sc.n_leaves; i += wg) l_leaves[i] = sc.leaves[i];
        __syncthreads();
    }
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(lds_raw);     // low 32 bits of the flat address = LDS offset
    const uint32_t lds_inner = lds_base + kSceneBase * 16u, lds_hot = lds_base + hot_base * 16u, lds_leaf = lds_base + leaf_base * 16u;
    auto fetch_tri = [&](int i) -> TriTest {
        uint4 a, b; uint32_t c;
        if (LDS_SCENE) {
            const uint32_t q = lds_hot + __umul24((uint32_t)i, 48u);
            a = lds_load4(q); b = lds_load4(q + 16); c = lds_load1(q + 32);
        } else {
            const uint4 *q = reinterpret_cast<const uint4 *>(sc.tri_hot + i);
            a = q[0]; b = q[1]; c = q[2].x;
        }
        TriTest t;
        t.v0 = mk3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
        t.e1 = mk3(__uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y));
        t.e2 = mk3(__uint_as_float(b.z), __uint_as_float(b.w), __uint_as_float(c));
        return t;
    };
    auto fetch_children = [&](uint32_t index) -> ChildPair {
        uint4 a, b, c; uint2 r;
        if (LDS_SCENE) {
            const uint32_t q = lds_inner + index * 64u;
            a = lds_load4(q); b = lds_load4(q + 16); c = lds_load4(q + 32); r = lds_load2(q + 48);
        } else {
            const uint4 *q = reinterpret_cast<const uint4 *>(sc.inner + index);
            a = q[0]; b = q[1]; c = q[2];
            r = *reinterpret_cast<const uint2 *>(&q[3]);
        }
        ChildPair p;
        p.min1 = mk3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
        p.max1 = mk3(__uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y));
        p.min2 = mk3(__uint_as_float(b.z), __uint_as_float(b.w), __uint_as_float(c.x));
        p.max2 = mk3(__uint_as_float(c.y), __uint_as_float(c.z), __uint_as_float(c.w));
        p.ref1 = r.x; p.ref2 = r.y;
        return p;
    };
    auto fetch_face_normal = [&](int prim) -> f3 {
        if (LDS_SCENE) {
            const uint32_t q = lds_hot + __umul24((uint32_t)prim, 48u) + 36u;
            return mk3(__uint_as_float(lds_load1(q)), __uint_as_float(lds_load1(q + 4)), __uint_as_float(lds_load1(q + 8)));
        }
        return ld3(sc.tri_hot[prim].fn);
    };
    // TriCold / MatDev / TexDev stay in HBM (L1-resident in practice): staging them in LDS too was measured -- no gain on
    // cornell, and it pushed room's footprint from 4 to 3 workgroups per CU (-18 %).
    auto fetch_cold = [&](int prim) -> TriCold { return sc.tri_cold[prim]; };
    auto fetch_mat = [&](int index) -> MatDev { return sc.mats[index]; };
    auto fetch_tex = [&](int index) -> TexDev { return sc.texs[index]; };
    auto fetch_leaf = [&](uint32_t id) -> LeafRange {
        if (LDS_SCENE) { const uint2 v = lds_load2(lds_leaf + id * 8u); LeafRange l; l.start = (int)v.x; l.count = (int)v.y; return l; }
        return sc.leaves[id];
    };

    int vote_node = fp.vote_node, vote_shade = fp.vote_shade;
    const bool leaf_chain = !GENERAL && fp.leaf_chain != 0;          // wave-uniform
    const int vote_dir = fp.vote_dir;
    const bool debug = GENERAL && fp.render_mode == 1;
    const bool sun = SUN || (GENERAL && fp.enable_sunlight && !debug);
    const f3 root_min = ld3(sc.root_min), root_max = ld3(sc.root_max);
    const uint32_t local_pixels = fp.width * fp.local_rows;

    // ---- lane state ----
    int stage = kNeedSample;
    int sp = 0;                      // stack height
    int cur = 0, end = 0;            // leaf cursor: triangles [cur, end) still to test
    bool shadow = false;             // current traversal is RayTest (BVHTraversal.cuh:76-134)
    bool occluded = false;
    Ray ray = make_ray(mk3(0, 0, 0), mk3(0, 0, 1));
    f3 hit_tuv = mk3(FLT_MAX, 0, 0);                                     // closest hit so far: distance and barycentrics
    float &hit_t = hit_tuv.x, &hit_u = hit_tuv.y, &hit_v = hit_tuv.z;
    int hit_prim = -1;
    float heat = 0;
    f3 light = mk3(0, 0, 0), throughput = mk3(1, 1, 1);
    // Origin and normal of the next bounce, kept from the shaded hit until its direction is drawn.  Only a sun shadow
    // traversal still needs the ray in between, so the kernels without one keep them in the dead ray's registers.
    f3 bounce_origin_own = mk3(0, 0, 0), bounce_normal_own = mk3(0, 0, 0);
    // The lean sunlight kernels need the ray for the shadow traversal but no longer the hit record: the origin goes into
    // the hit's three floats and the normal is re-derived from hit_prim (complemented when the face normal was flipped).
    f3 &bounce_origin = SUN ? hit_tuv : (GENERAL ? bounce_origin_own : ray.orig);
    f3 &bounce_normal = SHADOWS ? bounce_normal_own : ray.dir;
    f2 tex_uv; tex_uv.x = 0; tex_uv.y = 1;
    uint32_t seed = 0, slot = 0;     // slot: where this sample's colour goes in `samples`
    int bounce = 0;
    // bounce direction: 0 not requested, < 0 ready, > 0 pending (R steps work on it) = 1 + candidates drawn so far
    // (the count feeds the RNG cycle guard, device_math.hpp)
    int spec = 0;
    uint32_t spec_seed = 0;          // RNG state of the direction's draws (becomes `seed` when the direction is used)
    f3 spec_p = mk3(0, 0, 0);
    // wave-uniform sample pool: chunk = (tile, frame), pool_next = next unassigned sample of the chunk
    uint32_t chunk = 0, pool_next = 64;
    bool exhausted = false;
    // (Reading the queue head one chunk ahead was tried: vector-memory results return in order, so the next texel or
    // material load waits for the outstanding atomic anyway -- 2 % slower.)
    unsigned long long c_samples = 0, c_rays = 0, c_nodes = 0, c_inner = 0, c_tris = 0, c_htex = 0, c_hflat = 0,
                       c_srays = 0, c_sinner = 0, c_stris = 0;
    unsigned long long d_exec[4] = { 0, 0, 0, 0 }, d_lanes[4] = { 0, 0, 0, 0 };     // phase runs / lanes served
    unsigned long long d_time[4] = { 0, 0, 0, 0 }, d_t0 = 0;                         // shader-clock ticks spent in each phase (counting build only)
    const unsigned long long d_kernel_t0 = COUNT ? __builtin_amdgcn_s_memtime() : 0;

    auto begin_closest = [&]() {                         // TraceRay.cu:15-20 + BVHTraversal.cuh:22-26
        hit_t = FLT_MAX; hit_prim = -1; heat = 0; shadow = false;
        cur = end = 0; sp = 0;
        if (COUNT) c_rays++;
        if (sc.root_ref != kNoNode) {
            StackEntry e; e.ref = sc.root_ref; e.dist = slab_intersect(root_min, root_max, ray);
            // Lean: the pop-time test of BVHTraversal.cuh:38 (-1 < dist < FLT_MAX) can only ever fail for the root -- every
            // other entry was pushed with 0 <= dist < closest -- so it is applied here, once per ray instead of once per pop.
            if (GENERAL || (-1.0f < e.dist && e.dist < FLT_MAX)) {
                stack_store(0, e);
                sp = 1;
            }
        }
    };

    // After launching the ray of bounce index `bounce`: its hit will need a direction drawn from seed + bounce
    // (RayGen.cuh:91 then :134), unless sunlight draws come first (then the direction is requested at the hit).
    auto arm_direction = [&]() {
        if (!debug && !sun && bounce < fp.bounce_limit) { spec = 1; spec_seed = seed + (uint32_t)bounce; }
        else spec = 0;
    };

    // Wave loop.  One trip = maybe S, then R steps, then N steps, then T steps; every inner loop stops as soon as
    // its own supply of lanes is low and another phase has enough lanes waiting (thresholds vote_*).
    for (;;) {
        // class masks (scalar): a lane is T if it has triangles, else N if it has stack entries, else R / S by stage
        unsigned long long m_t = ballot(cur < end), m_sp = ballot(sp > 0);
        // m_dir: lanes BLOCKED on their bounce direction (hit shaded, direction not ready yet); m_pend: direction pending
        unsigned long long m_dir = ballot(stage == kNeedDir && spec >= 0), m_fin = ballot(stage == kFinished);
        unsigned long long m_pend = ballot(spec > 0);
        unsigned long long m_n = ~m_t & m_sp, m_idle = ~m_t & ~m_sp;
        unsigned long long m_r = m_idle & m_dir, m_s = m_idle & ~m_dir & ~m_fin;
        if ((m_t | m_n | m_r | m_s) == 0) break;

        // ================= S: shade, finish paths, deal samples, generate primary rays =================
        const int n_s = __popcll(m_s);
        if (n_s >= vote_shade || (m_t == 0 && m_n == 0 && n_s > 0 && n_s >= __popcll(m_r))) {
            if (COUNT) { d_exec[2]++; d_lanes[2] += (unsigned long long)n_s; d_t0 = __builtin_amdgcn_s_memtime(); }
            // S is the long, memory-latency-bound phase (material / texel chains): a wave inside it goes first, so its loads
            // are issued early and it is back in the compute phases sooner (+1-2 %, tools/ab_libs.py)
            __builtin_amdgcn_s_setprio(DRT_PRIO_S);
            const bool in_s = !(cur < end) && !(sp > 0) && !(stage == kNeedDir && spec >= 0) && stage != kFinished;
            // Lean paths gather light only where they end (the sky term below), in the S run that also stores the
            // sample: the running sum is zero on entry, and resetting it here frees its registers between S runs.
            if (!SHADOWS) light = mk3(0, 0, 0);
            // (a) a closest-hit traversal finished: RayGen.cuh:90-134
            if (in_s && stage == kTraceDone) {
                seed += (uint32_t)bounce;                                                  // :91
                bool path_done = false;
                if (hit_prim < 0) {                                                        // :99-108
                    if (debug && fp.debug_mode == 4) light = mk3(heat, heat, heat);
                    else light = light + sky_model(ray.dir, ld3(fp.sky_color)) * throughput * fp.sky_intensity;
                    path_done = true;
                } else {
                    const f3 uvw = mk3(1.0f - hit_u - hit_v, hit_u, hit_v);                // Intersection.cu:31
                    f3 position, normal;                                                   // ClosestHit.cuh:13-24
                    const bool front_face = closest_hit_frame(ray, hit_t, fetch_face_normal(hit_prim), position, normal);
                    const TriCold cold = fetch_cold(hit_prim);                             // :111-118
                    const MatDev mat = fetch_mat(cold.material);
                    // opt-in material model (drt.h drt_material_model; not reference behaviour, general kernel only): emission seen
                    // through the path so far, before this hit's albedo
                    if (GENERAL && fp.ext_emissive && !debug)
                        light = light + (ld3(sc.mats_ext[cold.material].emissive) * fp.ext_emissive_scale) * throughput;
                    if (mat.tex < 0) {
                        throughput = throughput * ld3(mat.albedo);
                        if (COUNT) c_hflat++;
                    } else {
                        tex_uv = interp_uv(cold, uvw);
                        throughput = throughput * tex_get_pixel(sc, fetch_tex(mat.tex), tex_uv);
                        if (COUNT) c_htex++;
                    }
                    bounce_origin = position + (normal * 0.001f);                          // :121
                    if (SUN) { if (!front_face) hit_prim = ~hit_prim; }                    // normal = +-face normal of hit_prim: re-derived at launch
                    else bounce_normal = normal;
                    if (GENERAL && fp.ext_specular && !debug) {
                        // opt-in: a Metallic material reflects; the mirror direction takes the normal's place until the fuzz is drawn,
                        // the roughness rides in tex_uv (only the UV debug view reads that) and the normal is re-derived from hit_prim
                        const MatExt ext = sc.mats_ext[cold.material];
                        tex_uv.x = 0;
                        if (ext.metallic) {
                            const f3 v = normalize(ray.dir);
                            bounce_normal = v - normal * (2.0f * dot(v, normal));
                            tex_uv.x = 1; tex_uv.y = ext.roughness;
                            if (!front_face) hit_prim = ~hit_prim;
                        }
                    }
                    stage = kShadowDone;                                                   // (b) below, now or after the shadow ray
                    occluded = true;
                    if (sun) {                                                             // :124-128
                        ray = make_ray(bounce_origin, ld3(fp.sunpos) + random_unit_vec3(seed) * 1.5f);
                        if (COUNT) c_srays++;
                        shadow = true; occluded = false; cur = end = 0; sp = 0;
                        if (sc.root_ref != kNoNode && !(slab_intersect(root_min, root_max, ray) < 0)) {   // BVHTraversal.cuh:95-103
                            StackEntry e; e.ref = sc.root_ref; e.dist = 0;
                            stack_store(0, e);
                            sp = 1;
                        }
                    }
                    if (debug) {                                                           // :137-161 (the bounce draw before it has no visible effect)
                        switch (fp.debug_mode) {
                        case 0: light = throughput; break;
                        case 1: light = normal; break;
                        case 2: light = uvw; break;
                        case 3: light = mk3(tex_uv.x, tex_uv.y, 0); break;
                        case 4: light = mk3(0, 0.1f, 0.1f) + mk3(heat, heat, heat); break;
                        default: break;
                        }
                        path_done = true;
                    }
                }
                if (path_done) { stage = kPathDone; sp = 0; cur = end = 0; spec = 0; }
            }
            // (b) the sun shadow traversal (if any) is over: add sunlight, ask for a bounce direction  RayGen.cuh:126-134
            if (in_s && stage == kShadowDone && sp == 0) {
                if (sun && !occluded) light = light + ld3(fp.suncol) * throughput;
                ++bounce;
                // :88 loop condition.  The bounce direction (randomUnitSphereVec3's rejection loop) is drawn one
                // candidate per R step; after the last bounce the reference still draws one, which nothing reads.
                if (bounce <= fp.bounce_limit) {
                    stage = kNeedDir;
                    if (spec == 0) { spec = 1; spec_seed = seed; }              // not drawn in the background: request it now
                } else {
                    stage = kPathDone;
                    spec = 0;
                }
            }
            // (b2) direction ready: launch the bounce ray  RayGen.cuh:133-134
            if (in_s && stage == kNeedDir && spec < 0) {
                seed = spec_seed;
                bool absorbed = false;
                if (SUN) {
                    const f3 fn = fetch_face_normal(hit_prim < 0 ? ~hit_prim : hit_prim);
                    ray = make_ray(bounce_origin, (hit_prim < 0 ? (-1.f * fn) : fn) + spec_p);
                } else if (GENERAL && fp.ext_specular && !debug && tex_uv.x != 0) {
                    const f3 fn = fetch_face_normal(hit_prim < 0 ? ~hit_prim : hit_prim);
                    const f3 n = hit_prim < 0 ? (-1.f * fn) : fn;
                    const f3 dir = bounce_normal + spec_p * tex_uv.y;                      // reflect(...) + roughness * fuzz
                    absorbed = !(dot(dir, n) > 0.0f);                                      // scattered into the surface: the path ends
                    ray = make_ray(bounce_origin, dir);
                } else
                ray = make_ray(bounce_origin, bounce_normal + spec_p);
                if (absorbed) { stage = kPathDone; sp = 0; cur = end = 0; spec = 0; }
                else {
                    begin_closest();
                    stage = kTraceDone;
                    arm_direction();
                }
            }
            // (c) path finished: post-process and park the sample's colour  RayGen.cuh:165-171
            if (in_s && stage == kPathDone) {
                if (!debug || fp.debug_mode == 0) {
                    if (fp.tone_mapping) light = uncharted2_filmic(light, fp.exposure);     // wave-uniform: a scalar branch
                    if (fp.gamma_correction) light = gamma_correction(light);
                }
                samples[slot] = make_float4(light.x, light.y, light.z, 0.0f);
                stage = kNeedSample;
            }
            // (d) deal the samples of the wave's chunk; pull a new chunk from the queue when it is used up
            uint32_t my_k = 0, my_chunk = 0;
            bool got = false;
            {
                bool need = in_s && stage == kNeedSample;
                for (;;) {
                    const unsigned long long m = ballot(need);
                    if (m == 0) break;
                    if (pool_next >= 64) {
                        unsigned int c = 0;
                        if (!exhausted && lane == 0) c = atomicAdd(chunk_counter, 1u);
                        c = __builtin_amdgcn_readfirstlane(c);
                        if (exhausted || c >= n_chunks) {
                            // the queue is empty: no new paths will refill this wave's lanes.  From here on what matters is
                            // how long the wave's slowest path takes, not how full its phases run, so lanes stop waiting for
                            // company before they shade or pop
                            if (!exhausted) { vote_shade = min(vote_shade, fp.vote_tail_shade); vote_node = min(vote_node, fp.vote_tail_node); }
                            exhausted = true;
                            if (need) { stage = kFinished; need = false; }
                            break;
                        }
                        chunk = c; pool_next = 0;
                    }
                    const int rank = lane_rank(m);
                    const int avail = 64 - (int)pool_next;
                    if (need && rank < avail) { my_k = pool_next + (uint32_t)rank; my_chunk = chunk; got = true; need = false; }
                    pool_next += (uint32_t)min(__popcll(m), avail);
                }
            }
            // (e) primary ray of the new sample  RayGen.cuh:63-86
            if (got) {
#ifndef DRT_CHUNK_ORDER
#define DRT_CHUNK_ORDER 4
#endif
                // chunk -> (tile, frame).  4: frame-major, tile rows visited with a stride (fp.row_step, coprime to the number of
                // tile rows): at any moment the resident waves work on tile rows from all over the image, so the mix of long
                // paths (VALU work) and short ones (sky: latency) is the image's average from the first chunk to the last.
                // 0: tile-major raster; 1: frame-major raster; 5: tile-major with the row stride  (tools/ab_chunk_order.sh)
                const uint32_t n_tiles_all = n_chunks / fp.n_frames;
                uint32_t tile, f_rel;
                if (DRT_CHUNK_ORDER == 1 || DRT_CHUNK_ORDER == 4) { f_rel = my_chunk / n_tiles_all; tile = my_chunk - f_rel * n_tiles_all; }
                else { tile = my_chunk / fp.n_frames; f_rel = my_chunk - tile * fp.n_frames; }
                uint32_t ty = tile / tiles_x;
                const uint32_t tx = tile - ty * tiles_x;
                if (DRT_CHUNK_ORDER >= 4) ty = (ty * fp.row_step) % (n_tiles_all / tiles_x);
                const uint32_t x = tx * 8u + (my_k & 7u), ly = ty * 8u + (my_k >> 3);
                if (x < fp.width && ly < fp.local_rows) {
                    const uint32_t y = ((ly / fp.stripe_rows) * fp.world + fp.rank) * fp.stripe_rows + (ly % fp.stripe_rows);
                    slot = f_rel * local_pixels + ly * fp.width + x;
                    f2 screen_uv;
                    screen_uv.x = ((float)x / (float)fp.width) * 2 - 1;
                    screen_uv.y = ((float)y / (float)fp.height) * 2 - 1;
                    seed = x + y * fp.width;
                    seed *= fp.frame_first + f_rel;
                    ray = camera_get_ray(fp, screen_uv, seed);
                    light = mk3(0, 0, 0); throughput = mk3(1, 1, 1);
                    tex_uv.x = 0; tex_uv.y = 1;
                    bounce = 0;
                    if (COUNT) c_samples++;
                    if (fp.bounce_limit >= 0) {
                        begin_closest();
                        stage = kTraceDone;
                        arm_direction();
                    } else {
                        stage = kPathDone;           // RayGen.cuh:88: the loop body never runs, the sample is black (post-processed)
                        sp = 0; cur = end = 0; spec = 0;
                    }
                }
                // a sample outside the image (partial tile) leaves the lane in kNeedSample: it asks again next time
            }
            m_sp = ballot(sp > 0);
            m_dir = ballot(stage == kNeedDir && spec >= 0); m_fin = ballot(stage == kFinished);
            m_pend = ballot(spec > 0);
            __builtin_amdgcn_s_setprio(0);
            if (COUNT) d_time[2] += __builtin_amdgcn_s_memtime() - d_t0;
        }

        // ================= R: one candidate of the bounce direction for every lane that has one pending
        //                    (Random.cu:50-58, drawn ahead of RayGen.cuh:133-134) =================
        for (;;) {
            const int n_pend = __popcll(m_pend);
            if (n_pend == 0) break;
            const int n_block = __popcll(~m_t & ~m_sp & m_dir);
            // run when enough directions are pending, when enough lanes are blocked on theirs, or when nothing else can run
            if (n_pend < fp.vote_spec && n_block < vote_dir && !(n_block > 0 && (m_t | (~m_t & m_sp)) == 0)) break;
            if (COUNT) { d_exec[3]++; d_lanes[3] += (unsigned long long)n_pend; d_t0 = __builtin_amdgcn_s_memtime(); }
            if (spec > 0) {
                f3 p;
                const bool accepted = random_unit_sphere_try(spec_seed, p);
                if (accepted || ++spec > kMaxTries) { spec_p = p; spec = -1; }
            }
            m_pend = ballot(spec > 0);
            m_dir = ballot(stage == kNeedDir && spec >= 0);
            if (COUNT) d_time[3] += __builtin_amdgcn_s_memtime() - d_t0;
        }
        // lanes whose direction just became ready are S lanes now; they are picked up by the next trip's S vote

        // ================= N: pop one stack entry per waiting lane (BVHTraversal.cuh:33-72 / :91-131) =================
        for (;;) {
            m_n = ~m_t & m_sp;
            const int n_n = __popcll(m_n);
            if (n_n == 0) break;
            if (n_n < vote_node && m_t != 0) break;
            if (COUNT) { d_exec[1]++; d_lanes[1] += (unsigned long long)n_n; d_t0 = __builtin_amdgcn_s_memtime(); }

            if (!(cur < end) && sp > 0) {
                --sp;
                const StackEntry e = stack_load(sp);
                bool visit = true;
                if (!GENERAL) {
                    // :41 (without a hit, hit_t = FLT_MAX > dist); :38 was applied to the root; a shadow traversal has no cull
                    visit = !((SUN && shadow ? FLT_MAX : hit_t) < e.dist);
                } else if (!shadow) {
                    if (!(-1.0f < e.dist && e.dist < FLT_MAX)) visit = false;                   // :38
                    else if (hit_prim >= 0 && hit_t < e.dist) visit = false;                    // :41
                }
                if (visit) {
                    if (GENERAL && !shadow) { heat += 0.05f; if (COUNT) c_nodes++; }            // :43
                    if (e.ref & kLeafBit) {
                        const LeafRange leaf = fetch_leaf(e.ref & ~kLeafBit);
                        cur = leaf.start; end = leaf.start + leaf.count;
                    } else if (!GENERAL) {
                        const ChildPair c = fetch_children(e.ref);
                        const float d1 = slab_entry_or_inf(c.min1, c.max1, ray);     // +inf = missed (device_math.hpp)
                        const float d2 = slab_entry_or_inf(c.min2, c.max2, ray);
                        const bool first_is_1 = d1 > d2;          // farther child first; child 2 first on ties (:63-70)
                        StackEntry ea, eb;                        // sort the two entries, then test each for the push
                        ea.ref = first_is_1 ? c.ref1 : c.ref2; ea.dist = first_is_1 ? d1 : d2;
                        eb.ref = first_is_1 ? c.ref2 : c.ref1; eb.dist = first_is_1 ? d2 : d1;
                        const float limit = SUN && shadow ? FLT_MAX : hit_t;        // RayTest pushes every box it hits (:122-129)
                        if (ea.dist < limit) { stack_store(sp, ea); ++sp; }
                        if (eb.dist < limit) {
                            // A near child that is a leaf goes straight to T: pushed, it would be this lane's next pop, and it would
                            // pass :41 because nothing changes hit_t in between (0 to -5 % on the benchmark scenes)
                            if (eb.ref & kLeafBit) {
                                const LeafRange leaf = fetch_leaf(eb.ref & ~kLeafBit);
                                cur = leaf.start; end = leaf.start + leaf.count;
                            } else { stack_store(sp, eb); ++sp; }
                        }
                    } else {
                        const ChildPair c = fetch_children(e.ref);
                        const float d1 = slab_intersect(c.min1, c.max1, ray);
                        const float d2 = slab_intersect(c.min2, c.max2, ray);
                        bool push1, push2;
                        if (GENERAL && shadow) { push1 = d1 >= 0; push2 = d2 >= 0; if (COUNT) c_sinner++; }   // :122-129
                        else { push1 = d1 >= 0 && d1 < hit_t; push2 = d2 >= 0 && d2 < hit_t; if (COUNT) c_inner++; }   // :63-70
                        const bool first_is_1 = d1 > d2;          // farther child first; child 2 first on ties
                        StackEntry e1; e1.ref = c.ref1; e1.dist = d1;
                        StackEntry e2; e2.ref = c.ref2; e2.dist = d2;
                        const StackEntry ea = first_is_1 ? e1 : e2, eb = first_is_1 ? e2 : e1;
                        const bool pa = first_is_1 ? push1 : push2, pb = first_is_1 ? push2 : push1;
                        if (pa) { stack_store(sp, ea); ++sp; }
                        if (pb) { stack_store(sp, eb); ++sp; }
                    }
                }
            }
            m_t = ballot(cur < end); m_sp = ballot(sp > 0);
            if (COUNT) d_time[1] += __builtin_amdgcn_s_memtime() - d_t0;
        }

        // ================= T: one triangle per lane (Intersection.cu:4-36, BVHTraversal.cuh:46-57 / :105-117) ==========
        for (;;) {
            if (m_t == 0) break;
            const unsigned long long idle = ~m_t & ~m_sp;
            if (__popcll(~m_t & m_sp) >= vote_node || __popcll(idle & m_dir) >= vote_dir ||
                __popcll(idle & ~m_dir & ~m_fin) >= vote_shade) break;
            if (COUNT) { d_exec[0]++; d_lanes[0] += (unsigned long long)__popcll(m_t); d_t0 = __builtin_amdgcn_s_memtime(); }
            constexpr int kTrisWide = TRIS;
            if (!GENERAL && kTrisWide > 2) {
                // three triangles of the leaf per step: more loads in flight per memory round trip, for trees read from HBM
                // (at 94 VGPR = one wave per SIMD less; launch_one decides where that trade pays)
                if (cur < end) {
                    const int i0 = cur;
                    cur = min(i0 + kTrisWide, end);
                    int idx[kTrisWide];
                    TriTest tt[kTrisWide];
#pragma unroll
                    for (int k = 0; k < kTrisWide; k++) { idx[k] = min(i0 + k, end - 1); tt[k] = fetch_tri(idx[k]); }
                    float tk[kTrisWide], uk[kTrisWide], vk[kTrisWide];
                    bool hk[kTrisWide];
#pragma unroll
                    for (int k = 0; k < kTrisWide; k++)
                        hk[k] = tri_intersect_flat(ray, tt[k].v0, tt[k].e1, tt[k].e2, tk[k], uk[k], vk[k]) & (i0 + k < end);
                    if (SUN && shadow) {
                        bool occ = false;
#pragma unroll
                        for (int k = 0; k < kTrisWide; k++) occ = occ || (hk[k] && (!ALPHA || any_hit(sc, idx[k], mk3(1.0f - uk[k] - vk[k], uk[k], vk[k]))));
                        if (occ) { occluded = true; cur = end = 0; sp = 0; }
                    } else {
#pragma unroll
                        for (int k = 0; k < kTrisWide; k++)
                            if (hk[k] && tk[k] < hit_t && (!ALPHA || any_hit(sc, idx[k], mk3(1.0f - uk[k] - vk[k], uk[k], vk[k])))) {
                                hit_t = tk[k]; hit_prim = idx[k]; hit_u = uk[k]; hit_v = vk[k];
                            }
                    }
                }
            } else if (!GENERAL && kTrianglesPerStep == 2) {
                // lean variant: two consecutive triangles of the leaf per step (both loads in flight together, two
                // independent dependency chains to interleave); hits are applied in leaf order, so ties resolve as in
                // the one-at-a-time loop.  A lane with one triangle left tests it twice and ignores the second result.
                if (cur < end) {
                    const int i = cur;
                    const bool two = i + 1 < end;
                    const int j = two ? i + 1 : i;
                    cur = j + 1;
                    const TriTest ta = fetch_tri(i), tb = fetch_tri(j);
                    float t0, u0, v0, t1, u1, v1;
                    const bool h0 = tri_intersect_flat(ray, ta.v0, ta.e1, ta.e2, t0, u0, v0);
                    const bool h1 = tri_intersect_flat(ray, tb.v0, tb.e1, tb.e2, t1, u1, v1) & two;
                    // (with RGBA textures a candidate also has to pass the alpha test: BVHTraversal.cuh:50-52, AnyHit.cuh:8-28)
                    if (SUN && shadow) {                                      // RayTest: any accepted hit ends the traversal (:105-117)
                        const bool occ = (h0 && (!ALPHA || any_hit(sc, i, mk3(1.0f - u0 - v0, u0, v0)))) ||
                                         (h1 && (!ALPHA || any_hit(sc, j, mk3(1.0f - u1 - v1, u1, v1))));
                        if (occ) { occluded = true; cur = end = 0; sp = 0; }
                    } else {
                        if (h0 && t0 < hit_t && (!ALPHA || any_hit(sc, i, mk3(1.0f - u0 - v0, u0, v0)))) { hit_t = t0; hit_prim = i; hit_u = u0; hit_v = v0; }
                        if (h1 && t1 < hit_t && (!ALPHA || any_hit(sc, j, mk3(1.0f - u1 - v1, u1, v1)))) { hit_t = t1; hit_prim = j; hit_u = u1; hit_v = v1; }
                    }
                }
                if (leaf_chain && !(cur < end) && sp > 0) {
                    // what the next N step would do for this lane if its top entry is a leaf that survives :41 -- done here, the
                    // lane stays in T.  (A culled leaf or an interior node is left to N.)  Pays on very shallow trees only
                    // (cornell -1.5 %, room +2 %): the host sets fp.leaf_chain by tree depth.
                    const StackEntry e = stack_load(sp - 1);
                    if ((e.ref & kLeafBit) && !((SUN && shadow ? FLT_MAX : hit_t) < e.dist)) {
                        --sp;
                        const LeafRange leaf = fetch_leaf(e.ref & ~kLeafBit);
                        cur = leaf.start; end = leaf.start + leaf.count;
                    }
                }
            } else if (cur < end) {
                const int i = cur++;
                const TriTest tri = fetch_tri(i);
                float t, u, v;
                const bool h = tri_intersect_flat(ray, tri.v0, tri.e1, tri.e2, t, u, v);
                if (GENERAL && shadow) {
                    if (COUNT) c_stris++;
                    if (h && any_hit(sc, i, mk3(1.0f - u - v, u, v))) { occluded = true; cur = end = 0; sp = 0; }
                } else {
                    if (COUNT) c_tris++;
                    if (h && t < hit_t) {
                        if (!GENERAL || any_hit(sc, i, mk3(1.0f - u - v, u, v))) { hit_t = t; hit_prim = i; hit_u = u; hit_v = v; }
                    }
                }
            }
            m_t = ballot(cur < end);
            if (SHADOWS || leaf_chain) m_sp = ballot(sp > 0);
            if (COUNT) d_time[0] += __builtin_amdgcn_s_memtime() - d_t0;
        }
    }

    if (fp.span && lane == 0) atomicMax(&fp.span[1], (unsigned long long)wall_clock64());
    if (COUNT && fp.counters) {
        atomicAdd(&fp.counters[0], c_samples); atomicAdd(&fp.counters[1], c_rays); atomicAdd(&fp.counters[2], c_nodes);
        atomicAdd(&fp.counters[3], c_inner); atomicAdd(&fp.counters[4], c_tris); atomicAdd(&fp.counters[5], c_htex);
        atomicAdd(&fp.counters[6], c_hflat); atomicAdd(&fp.counters[7], c_srays); atomicAdd(&fp.counters[8], c_sinner);
        atomicAdd(&fp.counters[9], c_stris);
        if (lane == 0)
            for (int k = 0; k < 4; k++) {
                atomicAdd(&fp.counters[10 + k], d_exec[k]); atomicAdd(&fp.counters[14 + k], d_lanes[k]);
                atomicAdd(&fp.counters[18 + k], d_time[k]);
            }
        if (lane == 0) atomicAdd(&fp.counters[22], __builtin_amdgcn_s_memtime() - d_kernel_t0);
    }
}

// RenderKernel.cu:29-34 for a batch of frames: per pixel, add the frames' colours in frame order, store the running sum
// and the resolved RGBA32F texel (sum / last frame index, alpha 1).
__global__ __launch_bounds__(256) void resolve_kernel(const float4 *samples, float *accum, float4 *rgba, uint32_t local_pixels,
                                                      uint32_t n_frames, uint32_t last_frame_index) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= local_pixels) return;
    f3 acc = ld3(accum + 3 * (size_t)p);
    for (uint32_t f = 0; f < n_frames; f++) {
        const float4 c = samples[(size_t)f * local_pixels + p];
        acc = acc + mk3(c.x, c.y, c.z);
    }
    accum[3 * (size_t)p + 0] = acc.x; accum[3 * (size_t)p + 1] = acc.y; accum[3 * (size_t)p + 2] = acc.z;
    const f3 out = acc / (float)last_frame_index;
    rgba[p] = make_float4(out.x, out.y, out.z, 1.0f);
}

// Debug: compares exact_rcp(x) with 1.0f / x for every float bit pattern in [first, first + count).
__global__ __launch_bounds__(256) void check_rcp_kernel(int which, uint32_t first, unsigned long long count, unsigned long long *mismatches,
                                                        unsigned long long *fast_path) {
    unsigned long long bad = 0, fast = 0;
    for (unsigned long long k = (unsigned long long)blockIdx.x * 256u + threadIdx.x; k < count; k += (unsigned long long)gridDim.x * 256u) {
        const float x = __uint_as_float(first + (uint32_t)k);
        const float a = which == 0 ? exact_rcp(x) : exact_sqrt(x), b = which == 0 ? 1.0f / x : sqrtf(x);
        const uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
        const bool both_nan = (a != a) && (b != b);
        if (ua != ub && !both_nan) bad++;
        const float ax = __builtin_fabsf(x);
        if (ax >= 0x1p-100f && ax <= 0x1p100f) fast++;
    }
    if (bad) atomicAdd(mismatches, bad);
    if (fast) atomicAdd(fast_path, fast);
}

// Debug: finds every 32-bit value that lies on a cycle of pcg_hash (Random.cu:6-11) of length <= max_len.
// out[0] = number found, then (value, cycle length) pairs.
__global__ __launch_bounds__(256) void hash_cycles_kernel(uint32_t max_len, uint32_t *out, uint32_t cap_pairs) {
    for (unsigned long long k = (unsigned long long)blockIdx.x * 256u + threadIdx.x; k < (1ull << 32); k += (unsigned long long)gridDim.x * 256u) {
        const uint32_t x = (uint32_t)k;
        uint32_t y = x;
        for (uint32_t len = 1; len <= max_len; len++) {
            y = pcg_hash(y);
            if (y == x) {
                const uint32_t slot = atomicAdd(out, 1u);
                if (slot < cap_pairs) { out[1 + 2 * slot] = x; out[2 + 2 * slot] = len; }
                break;
            }
        }
    }
}

// Occupancy of one kernel variant at a workgroup size: workgroups per CU (0 = does not fit)
template <int MODE, bool LDS_SCENE, bool REF16, int TRIS = 2>
int groups_per_cu(int threads, size_t lds) {
    auto kernel = wave_queue_kernel<MODE, LDS_SCENE, REF16, TRIS>;
    if (lds > 160 * 1024) return 0;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 0;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, lds) != hipSuccess || n < 1) return 0;
    return std::min(n, 8 * kThreads / threads);
}

template <int MODE, bool LDS_SCENE, bool REF16, int TRIS = 2>
hipError_t launch_config(const SceneView &sc, const FrameParams &fp, unsigned int *chunk_counter, float4 *samples, uint32_t stack_entries,
                         size_t lds_bytes, int threads, int per_cu, int num_cus, hipStream_t stream) {
    const uint32_t tiles_x = (fp.width + 7) / 8, tiles_y = (fp.local_rows + 7) / 8;
    auto kernel = wave_queue_kernel<MODE, LDS_SCENE, REF16, TRIS>;
    const int waves_per_wg = threads / 64;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    // persistent grid: as many workgroups as the chip keeps resident (registers and LDS decide), never more than
    // there are chunks to hand out.  Workgroups are independent, so a mis-estimate only costs speed.
    const uint64_t n_chunks = (uint64_t)tiles_x * tiles_y * fp.n_frames;
    if (n_chunks > 0xFFFFFFF0ull) return hipErrorInvalidValue;
    // When the caller keeps several launches in flight (drt_renderer_set_frames_in_flight), a launch with little work -- a
    // 1/8 shard of a 1080p frame is 32 400 chunks -- gets a smaller grid, about 16 chunks per wave: its waves live long
    // enough to amortise their fill and drain, and the other launches find free CU slots instead of queueing behind a
    // full-size grid (1/8 shard, 3 frames in flight: 0.60 -> 0.57 ms per step; a whole frame still gets every slot).
    // A launch that has the GPU to itself wants every slot it can fill: one chunk per wave.
    // (the grid never drops below this launch's fair share of the GPU, slots / frames in flight: tiny launches must not
    // leave the machine empty)
    static const uint64_t chunks_per_wg_env = std::getenv("DRT_CHUNKS_PER_WG") ? (uint64_t)std::max(1, std::atoi(std::getenv("DRT_CHUNKS_PER_WG"))) : 0;
    const uint64_t slots = (uint64_t)num_cus * per_cu;
    uint64_t want = std::min<uint64_t>(slots, (n_chunks + waves_per_wg - 1) / waves_per_wg);      // at least one chunk per wave
    if (chunks_per_wg_env) want = std::min<uint64_t>(want, (n_chunks + chunks_per_wg_env - 1) / chunks_per_wg_env);
    else if (fp.frames_in_flight > 1)
        want = std::min<uint64_t>(want, std::max<uint64_t>((n_chunks + 16 * waves_per_wg - 1) / (16 * waves_per_wg), slots / (uint64_t)fp.frames_in_flight));
    const int blocks = (int)std::max<uint64_t>(1, want);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds_bytes, stream, sc, fp, chunk_counter, (uint32_t)n_chunks, tiles_x, samples, stack_entries);
    return hipGetLastError();
}

// ---- packaging of a launch: workgroup size, stack entry size, triangles per T step ----
// Every packaging computes the same bits (tests/test_gpu_parity.py), so which one runs is a matter of time only, and a
// renderer launches the same kind of work over and over: the legal candidates of a (kernel, scene, view class) are each TIMED
// on the first launches that are big enough to tell (their execution span, drt_renderer_kernel_span) and the fastest is kept
// (WaveQueueCache, one per renderer -- nothing here is shared between renderers, devices or threads).
//  * Workgroup size.  Every workgroup stages its own copy of an LDS scene next to its lanes' stacks, so a scene of some size
//    (room: 17.6 KB) is amortised over twice the waves by a 512-thread group.  (Waves never synchronise after the staging.)
//  * Stack entry size.  A deep tree read from HBM is limited by its stacks alone (16 levels x 8 B x 256 lanes = 32 KB per
//    group: 5 per CU); 6-byte entries (16-bit references, when they fit) make it 6, at one more LDS operation per access.
//  * Triangles per T step.  Three at 94 VGPRs (one wave per SIMD less) for trees read from HBM: more loads in flight per
//    memory round trip.
// Round 1 chose among them by rules fitted to four views (camera inside the scene's bounds or not); what used to be the rule
// is now only the order in which the candidates are tried.
template <int MODE, bool LDS_SCENE>
std::vector<WqVariant> wave_queue_candidates(const SceneView &sc, uint32_t stack_entries, size_t scene_lds_bytes, bool camera_inside) {
    constexpr bool kLean = MODE == 0 || MODE == 3 || MODE == 4 || MODE == 5;
    static const bool only_small = std::getenv("DRT_WG_THREADS") && std::atoi(std::getenv("DRT_WG_THREADS")) == kThreads;      // A/B switches
    static const bool only_wide = std::getenv("DRT_STACK_REF16") && std::atoi(std::getenv("DRT_STACK_REF16")) == 0;
    static const bool wide_allowed = !(std::getenv("DRT_TRIS_WIDE") && std::atoi(std::getenv("DRT_TRIS_WIDE")) == 0);
    std::vector<WqVariant> out;
    auto add = [&](int threads, int entry, int tris, int per_cu) {
        if (per_cu < 1) return;
        if (const char *cap = std::getenv("DRT_MAX_BLOCKS_PER_CU")) per_cu = std::max(1, std::min(per_cu, std::atoi(cap)));
        out.push_back(WqVariant{ threads, entry, tris, per_cu });
    };
    const int plain = groups_per_cu<MODE, LDS_SCENE, false>(kThreads, (size_t)stack_entries * kThreads * 8 + scene_lds_bytes);
    add(kThreads, 8, 2, plain);
    if (LDS_SCENE && scene_lds_bytes >= 4096 && !only_small) {
        const int n = groups_per_cu<MODE, LDS_SCENE, false>(kBigThreads, (size_t)stack_entries * kBigThreads * 8 + scene_lds_bytes);
        if (n * kBigThreads > plain * kThreads) add(kBigThreads, 8, 2, n);                 // only when it keeps more waves resident
    }
    if (kLean && !LDS_SCENE && !only_wide && sc.n_inner < 0x8000u && sc.n_leaves < 0x8000u) {
        const int n = groups_per_cu<MODE, LDS_SCENE, kLean && !LDS_SCENE>(kThreads, (size_t)stack_entries * kThreads * 6 + scene_lds_bytes);
        if (n > plain) add(kThreads, 6, 2, n);
    }
    if (MODE == 0 && !LDS_SCENE && wide_allowed)
        add(kThreads, 8, 3, groups_per_cu<MODE, LDS_SCENE, false, (MODE == 0 && !LDS_SCENE) ? 3 : 2>(kThreads, (size_t)stack_entries * kThreads * 8 + scene_lds_bytes));
    if (out.empty()) out.push_back(WqVariant{ kThreads, 8, 2, 1 });
    // the order of the trials = round 1's rules: from inside, more waves first; from outside (sky around), more loads in flight first
    auto rank = [&](const WqVariant &v) { return camera_inside ? (v.entry_bytes == 6 ? 0 : (v.threads == kBigThreads ? 1 : (v.tris == 3 ? 3 : 2)))
                                                               : (v.tris == 3 ? 0 : (v.threads == kBigThreads ? 1 : (v.entry_bytes == 6 ? 3 : 2))); };
    std::stable_sort(out.begin(), out.end(), [&](const WqVariant &a, const WqVariant &b) { return rank(a) < rank(b); });
    return out;
}

template <int MODE, bool LDS_SCENE>
hipError_t launch_one(const SceneView &sc, const FrameParams &fp, unsigned int *chunk_counter, float4 *samples,
                      uint32_t stack_entries, size_t scene_lds_bytes, int num_cus, hipStream_t stream, int *launch_shape, WaveQueueCache &cache) {
    constexpr bool kLean = MODE == 0 || MODE == 3 || MODE == 4 || MODE == 5;
    bool camera_inside = true;
    for (int k = 0; k < 3; k++) camera_inside = camera_inside && fp.cam_pos[k] >= sc.root_min[k] && fp.cam_pos[k] <= sc.root_max[k];
    // one plan per (kernel, scene shape, view class)
    const uint64_t key = ((((uint64_t)scene_lds_bytes * 131u + stack_entries) * 131u + sc.n_tris) * 16u + (uint64_t)MODE * 2u + (LDS_SCENE ? 1u : 0u)) * 2u + (camera_inside ? 1u : 0u);
    const WqVariant v = measured_choice(cache, key, (double)fp.width * fp.local_rows * fp.n_frames,
                                        [&] { return wave_queue_candidates<MODE, LDS_SCENE>(sc, stack_entries, scene_lds_bytes, camera_inside); });
    const size_t lds_bytes = (size_t)stack_entries * v.threads * v.entry_bytes + scene_lds_bytes;
    if (launch_shape) { launch_shape[0] = (int)stack_entries; launch_shape[1] = v.per_cu; launch_shape[2] = (int)(lds_bytes / 1024); launch_shape[3] = v.threads + (v.entry_bytes == 6 ? 1 : 0) + (v.tris == 3 ? 2 : 0); }
    if (v.tris == 3)
        return launch_config<MODE, LDS_SCENE, false, (MODE == 0 && !LDS_SCENE) ? 3 : 2>(sc, fp, chunk_counter, samples, stack_entries, lds_bytes, v.threads, v.per_cu, num_cus, stream);
    if (v.entry_bytes == 6)
        return launch_config<MODE, LDS_SCENE, kLean && !LDS_SCENE>(sc, fp, chunk_counter, samples, stack_entries, lds_bytes, v.threads, v.per_cu, num_cus, stream);
    return launch_config<MODE, LDS_SCENE, false>(sc, fp, chunk_counter, samples, stack_entries, lds_bytes, v.threads, v.per_cu, num_cus, stream);
}

hipError_t launch_mode(const SceneView &sc, const FrameParams &fp, int mode, bool lds_scene, unsigned int *chunk_counter,
                       float4 *samples, uint32_t stack_entries, size_t scene_lds_bytes, int num_cus, hipStream_t stream, int *launch_shape,
                       WaveQueueCache &cache) {
    if (lds_scene) {
        if (mode == 0) return launch_one<0, true>(sc, fp, chunk_counter, samples, stack_entries, scene_lds_bytes, num_cus, stream, launch_shape, cache);
        if (mode == 1) return launch_one<1, true>(sc, fp, chunk_counter, samples, stack_entries, scene_lds_bytes, num_cus, stream, launch_shape, cache);
        if (mode == 3) return launch_one<3, true>(sc, fp, chunk_counter, samples, stack_entries, scene_lds_bytes, num_cus, stream, launch_shape, cache);
        if (mode == 4) return launch_one<4, true>(sc, fp, chunk_counter, samples, stack_entries, scene_lds_bytes, num_cus, stream, launch_shape, cache);
        if (mode == 5) return launch_one<5, true>(sc, fp, chunk_counter, samples, stack_entries, scene_lds_bytes, num_cus, stream, launch_shape, cache);
        return launch_one<2, true>(sc, fp, chunk_counter, samples, stack_entries, scene_lds_bytes, num_cus, stream, launch_shape, cache);
    }
    if (mode == 0) return launch_one<0, false>(sc, fp, chunk_counter, samples, stack_entries, scene_lds_bytes, num_cus, stream, launch_shape, cache);
    if (mode == 1) return launch_one<1, false>(sc, fp, chunk_counter, samples, stack_entries, scene_lds_bytes, num_cus, stream, launch_shape, cache);
    if (mode == 3) return launch_one<3, false>(sc, fp, chunk_counter, samples, stack_entries, scene_lds_bytes, num_cus, stream, launch_shape, cache);
    if (mode == 4) return launch_one<4, false>(sc, fp, chunk_counter, samples, stack_entries, scene_lds_bytes, num_cus, stream, launch_shape, cache);
    if (mode == 5) return launch_one<5, false>(sc, fp, chunk_counter, samples, stack_entries, scene_lds_bytes, num_cus, stream, launch_shape, cache);
    return launch_one<2, false>(sc, fp, chunk_counter, samples, stack_entries, scene_lds_bytes, num_cus, stream, launch_shape, cache);
}

}  // namespace

// The plan of `key` (made from `candidates()` the first time) and the candidate this launch uses: the chosen one, or -- while the
// plan is still measuring -- the one with the fewest trials so far (a batch keeps one candidate; wave_queue_report feeds the time back).
WqVariant measured_choice(WaveQueueCache &cache, uint64_t key, double samples, const std::function<std::vector<WqVariant>()> &candidates) {
    WqPlan *plan = nullptr;
    for (WqPlan &p : cache.plans) if (p.key == key) plan = &p;
    if (!plan) {
        if (cache.plans.size() >= 32) cache.plans.erase(cache.plans.begin());
        cache.plans.emplace_back();
        plan = &cache.plans.back();
        plan->key = key;
        plan->cands = candidates();
        plan->ns_per_sample.assign(plan->cands.size(), -1.0);
        plan->trials.assign(plan->cands.size(), 0);
        plan->chosen = plan->cands.size() == 1 ? 0 : -1;
    }
    int use = plan->chosen;
    if (use < 0) {
        if (cache.batch_key == key && cache.batch_cand >= 0) use = cache.batch_cand;
        else { use = 0; for (size_t i = 1; i < plan->cands.size(); i++) if (plan->trials[i] < plan->trials[(size_t)use]) use = (int)i; }
    }
    cache.batch_key = key; cache.batch_cand = use;
    cache.batch_samples += samples;
    return plan->cands[(size_t)use];
}


hipError_t launch_check_rcp(int which, uint32_t first_bits, unsigned long long count, unsigned long long *d_out2, hipStream_t stream) {
    hipLaunchKernelGGL(check_rcp_kernel, dim3(4096), dim3(256), 0, stream, which, first_bits, count, d_out2, d_out2 + 1);
    return hipGetLastError();
}

hipError_t launch_hash_cycles(uint32_t max_len, uint32_t *d_out, uint32_t cap_pairs, hipStream_t stream) {
    hipLaunchKernelGGL(hash_cycles_kernel, dim3(8192), dim3(256), 0, stream, max_len, d_out, cap_pairs);
    return hipGetLastError();
}

// ordered accumulate + resolve of the samples of one launch (RenderKernel.cu:29-34), shared by the tracing kernels
hipError_t launch_resolve(const FrameParams &fp, void *samples, hipStream_t stream) {
    const uint32_t local_pixels = fp.width * fp.local_rows;
    hipLaunchKernelGGL(resolve_kernel, dim3((local_pixels + 255) / 256), dim3(256), 0, stream, static_cast<const float4 *>(samples), fp.accum,
                       reinterpret_cast<float4 *>(fp.rgba), local_pixels, fp.n_frames, fp.frame_first + fp.n_frames - 1);
    return hipGetLastError();
}

size_t wave_queue_scene_lds_bytes(const SceneView &sc) {
    return (size_t)sc.n_inner * sizeof(InnerNode) + (size_t)sc.n_tris * sizeof(TriHot) + (((size_t)sc.n_leaves * sizeof(LeafRange) + 15) & ~(size_t)15);
}

size_t wave_queue_sample_bytes(const FrameParams &fp) {
    return (size_t)fp.width * fp.local_rows * fp.n_frames * sizeof(float4);
}

// A batch is over and its tracing kernels took span_ms on the device: feed the measurement to the plan that is still choosing.
void wave_queue_report(WaveQueueCache &cache, float span_ms) {
    const uint64_t key = cache.batch_key;
    const int cand = cache.batch_cand;
    const double samples = cache.batch_samples;
    cache.batch_cand = -1; cache.batch_samples = 0;
    if (cand < 0 || span_ms <= 0.f || samples < 262144.0) return;          // too small a launch says nothing about steady-state speed
    for (WqPlan &p : cache.plans) {
        if (p.key != key || p.chosen >= 0 || (size_t)cand >= p.cands.size()) continue;
        const double t = (double)span_ms * 1e6 / samples;
        double &best = p.ns_per_sample[(size_t)cand];
        best = best < 0 ? t : std::min(best, t);
        p.trials[(size_t)cand]++;
        bool done = true;
        for (int n : p.trials) done = done && n >= 2;                       // two timed launches each
        if (done) {
            p.chosen = 0;
            for (size_t i = 1; i < p.cands.size(); i++) if (p.ns_per_sample[i] < p.ns_per_sample[(size_t)p.chosen]) p.chosen = (int)i;
        }
    }
}

hipError_t launch_wave_queue(const SceneView &sc, const FrameParams &fp, int bvh_depth, int mode, bool scene_has_alpha,
                             unsigned int *chunk_counter, void *samples, int num_cus, hipStream_t stream, const char **kernel_name,
                             int *launch_shape, WaveQueueCache &cache) {
    if (fp.width == 0 || fp.local_rows == 0 || fp.n_frames == 0) return hipSuccess;
    if (mode == 0 && fp.render_mode != 0) mode = 1;                                    // debug views: the general kernel
    if (mode == 0) mode = fp.enable_sunlight ? (scene_has_alpha ? 5 : 4) : (scene_has_alpha ? 3 : 0);
    // one slot per BVH level is all a depth-first walk that pushes both children can ever hold (BVHTraversal.cuh:20
    // fixes it at 64, which is also the reference's limit)
    if (bvh_depth > 64) return hipErrorInvalidValue;
    const int stack = std::max(bvh_depth, 1);
    const size_t scene_bytes = wave_queue_scene_lds_bytes(sc);
    static const size_t lds_scene_budget = std::getenv("DRT_LDS_SCENE_KB") ? (size_t)std::atoi(std::getenv("DRT_LDS_SCENE_KB")) * 1024 : kLdsSceneBytes;
    // (a small scene under a degenerate, very deep tree: the stacks of one 256-thread group and the scene copy must fit the
    // CU's 160 KB together, else the tree is read from HBM and the stacks have the LDS to themselves)
    const bool lds_scene = scene_bytes <= lds_scene_budget && scene_bytes + (size_t)stack * kThreads * sizeof(StackEntry) <= 160u * 1024u;
    hipError_t e = hipSuccess;                 // (*chunk_counter is zero: drt_capi.cpp hands out zeroed counters)
    static const char *names[2][6] = { { "wave_queue<lean,hbm-scene>", "wave_queue<general,hbm-scene>", "wave_queue<counting,hbm-scene>", "wave_queue<lean+alpha,hbm-scene>",
                                         "wave_queue<lean+sun,hbm-scene>", "wave_queue<lean+alpha+sun,hbm-scene>" },
                                       { "wave_queue<lean,lds-scene>", "wave_queue<general,lds-scene>", "wave_queue<counting,lds-scene>", "wave_queue<lean+alpha,lds-scene>",
                                         "wave_queue<lean+sun,lds-scene>", "wave_queue<lean+alpha+sun,lds-scene>" } };
    if (kernel_name) *kernel_name = names[lds_scene ? 1 : 0][mode];
    float4 *s4 = static_cast<float4 *>(samples);
    // (the one-frame shortcut of path_pool -- accumulate and resolve inside the tracing kernel -- costs this kernel a register
    // too many: its 6-byte-stack variants go from 80 to 81 VGPRs = 6 -> 5 waves per SIMD)
    e = launch_mode(sc, fp, mode, lds_scene, chunk_counter, s4, (uint32_t)stack, lds_scene ? scene_bytes : 0, num_cus, stream, launch_shape, cache);
    if (e != hipSuccess) return e;
    return launch_resolve(fp, samples, stream);
}

}  // namespace drt
