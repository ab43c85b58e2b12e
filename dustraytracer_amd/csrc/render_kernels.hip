// render_kernels.hip -- the path-tracing megakernel for gfx950 (MI355X), hand-written HIP.
//
// Reference being replaced: __global__ kernel (Core/Kernel/RenderKernel.cu:20-35) and everything it
// inlines: RayGen (Shaders/RayGen.cuh:63-172), TraceRay/RayTest (Kernel/TraceRay.cu:15-38),
// traverseBVH/_raytest (BVH/BVHTraversal.cuh:14-134), Intersection, AnyHit, ClosestHit, Miss.
//
// What the shipped library keeps of this file: the device known-answer-test kernel and the rank-0 de-interleave pass.
// Kernel "pixel_walk" (round 1's first correct path) is compiled only with -DDRT_WITH_PIXEL_WALK -- `make pixel-walk` builds
// dustraytracer_amd/libdrt_hip_pixel_walk.so, which the tests load as a third, straightforward implementation to cross-check
// the production kernels; libdrt_hip.so does not contain it (DRT_KERNEL=pixel_walk is refused there).
// Kernel "pixel_walk":
//   * one lane per pixel, 8x8 pixel tile per wave64, 4 tiles (32x8 pixels) per 256-thread workgroup
//   * a lane keeps its pixel for all frames of a batch: the running sum lives in registers and the
//     framebuffer is touched once per batch (12 B read + 12 B + 16 B written per pixel)
//   * traversal stack (node reference + entry distance) in LDS, one bank per lane: entry [level][tid]
//     sits in bank tid % 32 whatever the level, so pushes and pops never conflict
//   * scene read through the SoA records of device_scene.hpp (interior visit = 4 x 16 B)
// No MFMA: this is branchy scalar fp32 / u32 work.
#include <hip/hip_runtime.h>

#include "device_math.hpp"
#include "device_scene.hpp"
#include "device_access.hpp"
#include "render_kernels.hpp"

namespace drt {

namespace {

#ifdef DRT_WITH_PIXEL_WALK
constexpr int kBlockThreads = 256;

struct Counters {
    unsigned long long samples = 0, rays = 0, node_visits = 0, inner_visits = 0, tri_tests = 0, hits_textured = 0,
                       hits_flat = 0, shadow_rays = 0, inner_visits_shadow = 0, tri_tests_shadow = 0;
};

struct Hit {
    float t;
    int prim;      // -1 = miss
    f3 uvw;
    float heat;    // Core/BVH/BVHTraversal.cuh:43 visit counter (payload.color, all three channels equal)
};

// LDS traversal stack, STACK levels x 256 lanes.
template <int STACK>
struct LdsStack {
    uint32_t (*ref)[kBlockThreads];
    float (*dist)[kBlockThreads];
    int tid;
};

// ---- BVH/BVHTraversal.cuh:14-73 ----
template <int STACK, bool COUNT>
DRT_DEV Hit traverse_closest(const SceneView &sc, const Ray &ray, const LdsStack<STACK> &st, Counters &cnt) {
    Hit hit;
    hit.t = FLT_MAX;                 // ray.interval.max (RayGen.cuh:78, TraceRay.cu:18)
    hit.prim = -1;
    hit.uvw = mk3(0, 0, 0);
    hit.heat = 0;
    if (sc.root_ref == kNoNode) return hit;
    int sp = 0;
    st.ref[0][st.tid] = sc.root_ref;
    st.dist[0][st.tid] = slab_intersect(ld3(sc.root_min), ld3(sc.root_max), ray);
    sp = 1;
    while (sp > 0) {
        --sp;
        uint32_t ref = st.ref[sp][st.tid];
        float node_dist = st.dist[sp][st.tid];
        if (!(-1.0f < node_dist && node_dist < FLT_MAX)) continue;                 // :38 interval (-1, FLT_MAX)
        if (hit.prim >= 0 && hit.t < node_dist) continue;                          // :41
        hit.heat += 0.05f;                                                         // :43
        if (COUNT) cnt.node_visits++;
        if (ref & kLeafBit) {
            LeafRange leaf = sc.leaves[ref & ~kLeafBit];
            for (int i = leaf.start; i < leaf.start + leaf.count; i++) {           // :46-57
                TriTest tri = load_tri(sc.tri_hot, i);
                float t; f3 uvw;
                if (COUNT) cnt.tri_tests++;
                if (tri_intersect(ray, tri.v0, tri.e1, tri.e2, t, uvw) && t < hit.t) {
                    if (!any_hit(sc, i, uvw)) continue;
                    hit.t = t; hit.prim = i; hit.uvw = uvw;
                }
            }
        } else {
            if (COUNT) cnt.inner_visits++;
            ChildPair c = load_children(sc.inner, ref);
            float d1 = slab_intersect(c.min1, c.max1, ray);
            float d2 = slab_intersect(c.min2, c.max2, ray);
            bool push1 = d1 >= 0 && d1 < hit.t, push2 = d2 >= 0 && d2 < hit.t;
            if (d1 > d2) {                                                         // :63-70 farther child first
                if (push1) { st.ref[sp][st.tid] = c.ref1; st.dist[sp][st.tid] = d1; ++sp; }
                if (push2) { st.ref[sp][st.tid] = c.ref2; st.dist[sp][st.tid] = d2; ++sp; }
            } else {
                if (push2) { st.ref[sp][st.tid] = c.ref2; st.dist[sp][st.tid] = d2; ++sp; }
                if (push1) { st.ref[sp][st.tid] = c.ref1; st.dist[sp][st.tid] = d1; ++sp; }
            }
        }
    }
    return hit;
}

// ---- BVH/BVHTraversal.cuh:76-134 ----
template <int STACK, bool COUNT>
DRT_DEV bool traverse_any(const SceneView &sc, const Ray &ray, const LdsStack<STACK> &st, Counters &cnt) {
    if (sc.root_ref == kNoNode) return false;
    if (slab_intersect(ld3(sc.root_min), ld3(sc.root_max), ray) < 0) return false;       // :95-103 (root only)
    int sp = 0;
    st.ref[0][st.tid] = sc.root_ref;
    sp = 1;
    while (sp > 0) {
        --sp;
        uint32_t ref = st.ref[sp][st.tid];
        if (ref & kLeafBit) {
            LeafRange leaf = sc.leaves[ref & ~kLeafBit];
            for (int i = leaf.start; i < leaf.start + leaf.count; i++) {
                TriTest tri = load_tri(sc.tri_hot, i);
                float t; f3 uvw;
                if (COUNT) cnt.tri_tests_shadow++;
                if (tri_intersect(ray, tri.v0, tri.e1, tri.e2, t, uvw) && any_hit(sc, i, uvw)) return true;
            }
        } else {
            if (COUNT) cnt.inner_visits_shadow++;
            ChildPair c = load_children(sc.inner, ref);
            float h1 = slab_intersect(c.min1, c.max1, ray);
            float h2 = slab_intersect(c.min2, c.max2, ray);
            if (h1 > h2) {                                                         // :122-129
                if (h1 >= 0) { st.ref[sp][st.tid] = c.ref1; ++sp; }
                if (h2 >= 0) { st.ref[sp][st.tid] = c.ref2; ++sp; }
            } else {
                if (h2 >= 0) { st.ref[sp][st.tid] = c.ref2; ++sp; }
                if (h1 >= 0) { st.ref[sp][st.tid] = c.ref1; ++sp; }
            }
        }
    }
    return false;
}

// ---- RayGen, Shaders/RayGen.cuh:63-172 ----
template <int STACK, bool COUNT>
DRT_DEV f3 ray_gen(const SceneView &sc, const FrameParams &fp, uint32_t x, uint32_t y, uint32_t frameidx,
                   const LdsStack<STACK> &st, Counters &cnt) {
    f2 screen_uv;
    screen_uv.x = ((float)x / (float)fp.width) * 2 - 1;                            // :65-66
    screen_uv.y = ((float)y / (float)fp.height) * 2 - 1;
    uint32_t seed = x + y * fp.width;                                              // :74-75
    seed *= frameidx;
    Ray ray = camera_get_ray(fp, screen_uv, seed);

    f3 light = mk3(0, 0, 0), throughput = mk3(1, 1, 1);
    f2 tex_uv; tex_uv.x = 0; tex_uv.y = 1;                                         // :85
    const bool debug = fp.render_mode == 1;
    if (COUNT) cnt.samples++;

    for (int i = 0; i <= fp.bounce_limit; i++) {                                   // :88
        if (COUNT) cnt.rays++;
        Hit hit = traverse_closest<STACK, COUNT>(sc, ray, st, cnt);                // TraceRay.cu:15-32
        seed += (uint32_t)i;                                                       // :91

        if (hit.prim < 0) {                                                        // :99-108 (Miss.cuh)
            if (fp.debug_mode == 4 && debug) {
                light = mk3(hit.heat, hit.heat, hit.heat);
            } else {
                f3 sky = sky_model(ray.dir, ld3(fp.sky_color));
                light = light + sky * throughput * fp.sky_intensity;
            }
            break;
        }

        // ClosestHit.cuh:4-28
        f3 position, normal;
        closest_hit_frame(ray, hit.t, ld3(sc.tri_hot[hit.prim].fn), position, normal);

        const TriCold cold = sc.tri_cold[hit.prim];                                // :111-118
        const MatDev mat = sc.mats[cold.material];
        if (mat.tex < 0) {
            throughput = throughput * ld3(mat.albedo);
            if (COUNT) cnt.hits_flat++;
        } else {
            tex_uv = interp_uv(cold, hit.uvw);
            throughput = throughput * tex_get_pixel(sc, sc.texs[mat.tex], tex_uv);
            if (COUNT) cnt.hits_textured++;
        }

        f3 new_origin = position + (normal * 0.001f);                              // :121

        if (fp.enable_sunlight && !debug) {                                        // :124-128
            Ray shadow = make_ray(new_origin, ld3(fp.sunpos) + random_unit_vec3(seed) * 1.5f);
            if (COUNT) cnt.shadow_rays++;
            if (!traverse_any<STACK, COUNT>(sc, shadow, st, cnt)) light = light + ld3(fp.suncol) * throughput;
        }

        ray = make_ray(new_origin, normal + random_unit_sphere_vec3(seed));        // :133-134

        if (debug) {                                                               // :137-161
            switch (fp.debug_mode) {
            case 0: light = throughput; break;
            case 1: light = normal; break;
            case 2: light = hit.uvw; break;
            case 3: light = mk3(tex_uv.x, tex_uv.y, 0); break;
            case 4: light = mk3(0, 0.1f, 0.1f) + mk3(hit.heat, hit.heat, hit.heat); break;
            default: break;
            }
            break;
        }
    }

    if (!debug || fp.debug_mode == 0) {                                            // :165-169
        if (fp.tone_mapping) light = uncharted2_filmic(light, fp.exposure);
        if (fp.gamma_correction) light = gamma_correction(light);
    }
    return light;
}

// ---- the kernel: RenderKernel.cu:20-35 over this device's rows, all frames of the batch ----
template <int STACK, bool COUNT>
__global__ __launch_bounds__(kBlockThreads) void pixel_walk_kernel(const SceneView sc, const FrameParams fp) {
    __shared__ uint32_t s_ref[STACK][kBlockThreads];
    __shared__ float s_dist[STACK][kBlockThreads];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    // 32x8 pixel block = four 8x8 wave tiles side by side
    const uint32_t x = blockIdx.x * 32u + (uint32_t)wave * 8u + (uint32_t)(lane & 7);
    const uint32_t ly = blockIdx.y * 8u + (uint32_t)(lane >> 3);
    if (x >= fp.width || ly >= fp.local_rows) return;
    // local row -> global row: local stripe s belongs to global stripe s*world + rank
    const uint32_t y = ((ly / fp.stripe_rows) * fp.world + fp.rank) * fp.stripe_rows + (ly % fp.stripe_rows);

    LdsStack<STACK> st{ s_ref, s_dist, tid };
    Counters cnt;
    const size_t p = (size_t)x + (size_t)ly * fp.width;
    f3 acc = ld3(fp.accum + 3 * p);
    uint32_t frame = fp.frame_first;
    for (uint32_t k = 0; k < fp.n_frames; k++, frame++) {
        f3 c = ray_gen<STACK, COUNT>(sc, fp, x, y, frame, st, cnt);
        acc = acc + c;                                                             // RenderKernel.cu:29
    }
    fp.accum[3 * p + 0] = acc.x; fp.accum[3 * p + 1] = acc.y; fp.accum[3 * p + 2] = acc.z;
    f3 out = acc / (float)(frame - 1);                                             // RenderKernel.cu:30
    reinterpret_cast<float4 *>(fp.rgba)[p] = make_float4(out.x, out.y, out.z, 1.0f);

    if (COUNT && fp.counters) {
        atomicAdd(&fp.counters[0], cnt.samples); atomicAdd(&fp.counters[1], cnt.rays);
        atomicAdd(&fp.counters[2], cnt.node_visits); atomicAdd(&fp.counters[3], cnt.inner_visits);
        atomicAdd(&fp.counters[4], cnt.tri_tests); atomicAdd(&fp.counters[5], cnt.hits_textured);
        atomicAdd(&fp.counters[6], cnt.hits_flat); atomicAdd(&fp.counters[7], cnt.shadow_rays);
        atomicAdd(&fp.counters[8], cnt.inner_visits_shadow); atomicAdd(&fp.counters[9], cnt.tri_tests_shadow);
    }
}

#endif  // DRT_WITH_PIXEL_WALK

// Known-answer tests of the device leaf functions (drt_debug_kat): the inputs/outputs are the ones of
// tests/golden/kat_ref.npz, which was produced by the reference's own compiled sources.
__global__ __launch_bounds__(256) void kat_kernel(int which, const uint32_t *in, uint32_t *out, uint32_t n, const FrameParams fp) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float *fin = reinterpret_cast<const float *>(in);
    float *fout = reinterpret_cast<float *>(out);
    if (which == 0 || which == 1) {             // in: seed -> out: vec3, seed [, tries]   (0 = unit vec, 1 = unit sphere)
        uint32_t seed = in[i];
        f3 p; uint32_t tries = 1;
        if (which == 0) p = random_unit_vec3(seed);
        else { for (;; tries++) { if (random_unit_sphere_try(seed, p) || tries >= (uint32_t)kMaxTries) break; } }
        fout[5 * i] = p.x; fout[5 * i + 1] = p.y; fout[5 * i + 2] = p.z; out[5 * i + 3] = seed; out[5 * i + 4] = tries;
    } else if (which == 2) {                    // in: orig3, dir3, min3, max3 -> out: slab distance
        const float *r = fin + 12 * i;
        const Ray ray = make_ray(mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5]));
        fout[i] = slab_intersect(mk3(r[6], r[7], r[8]), mk3(r[9], r[10], r[11]), ray);
    } else if (which == 3) {                    // in: orig3, dir3, v0, v1, v2 -> out: t, U, V, W, hit (straight-line test)
        const float *r = fin + 15 * i;
        const Ray ray = make_ray(mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5]));
        const f3 v0 = mk3(r[6], r[7], r[8]);
        const f3 e1 = mk3(r[9], r[10], r[11]) - v0, e2 = mk3(r[12], r[13], r[14]) - v0;
        float t, u, v;
        const bool h = tri_intersect_flat(ray, v0, e1, e2, t, u, v);
        fout[5 * i] = h ? t : -1.0f; fout[5 * i + 1] = h ? 1.0f - u - v : 0.f; fout[5 * i + 2] = h ? u : 0.f; fout[5 * i + 3] = h ? v : 0.f;
        out[5 * i + 4] = h ? 1u : 0u;
    } else if (which == 4) {                    // in: u, v, seed -> out: orig3, dir3, seed  (camera constants in fp)
        f2 uv; uv.x = fin[3 * i]; uv.y = fin[3 * i + 1];
        uint32_t seed = in[3 * i + 2];
        const Ray ray = camera_get_ray(fp, uv, seed);
        fout[7 * i] = ray.orig.x; fout[7 * i + 1] = ray.orig.y; fout[7 * i + 2] = ray.orig.z;
        fout[7 * i + 3] = ray.dir.x; fout[7 * i + 4] = ray.dir.y; fout[7 * i + 5] = ray.dir.z; out[7 * i + 6] = seed;
    } else if (which == 5) {                    // in: seed -> out: disk2, seed
        uint32_t seed = in[i];
        const f2 p = random_in_unit_disk(seed);
        fout[3 * i] = p.x; fout[3 * i + 1] = p.y; out[3 * i + 2] = seed;
    } else if (which == 6) {                    // in: orig3, dir3, t, face_normal3 -> out: position3, normal3, front_face
        const float *r = fin + 10 * i;
        const Ray ray = make_ray(mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5]));
        f3 position, normal;
        const bool front = closest_hit_frame(ray, r[6], mk3(r[7], r[8], r[9]), position, normal);
        fout[7 * i] = position.x; fout[7 * i + 1] = position.y; fout[7 * i + 2] = position.z;
        fout[7 * i + 3] = normal.x; fout[7 * i + 4] = normal.y; fout[7 * i + 5] = normal.z; out[7 * i + 6] = front ? 1u : 0u;
    }
}

// rank-0 side of the gather: shard r, local row ly  ->  image row y
__global__ void assemble_shards_kernel(const float4 *gathered, float4 *image, uint32_t width, uint32_t height,
                                       uint32_t stripe_rows, uint32_t world, uint32_t padded_rows) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t y = blockIdx.y;
    if (x >= width || y >= height) return;
    const uint32_t stripe = y / stripe_rows, rank = stripe % world, local_stripe = stripe / world;
    const uint32_t ly = local_stripe * stripe_rows + y % stripe_rows;
    image[(size_t)y * width + x] = gathered[((size_t)rank * padded_rows + ly) * width + x];
}

#ifdef DRT_WITH_PIXEL_WALK
template <int STACK>
hipError_t launch_stack(const SceneView &sc, const FrameParams &fp, bool count, hipStream_t stream) {
    dim3 grid((fp.width + 31) / 32, (fp.local_rows + 7) / 8), block(kBlockThreads);
    if (count) hipLaunchKernelGGL((pixel_walk_kernel<STACK, true>), grid, block, 0, stream, sc, fp);
    else hipLaunchKernelGGL((pixel_walk_kernel<STACK, false>), grid, block, 0, stream, sc, fp);
    return hipGetLastError();
}

#endif

}  // namespace

#ifdef DRT_WITH_PIXEL_WALK
bool pixel_walk_built_in() { return true; }
hipError_t launch_render(const SceneView &sc, const FrameParams &fp, int bvh_depth, bool count, hipStream_t stream,
                         const char **kernel_name) {
    if (fp.width == 0 || fp.local_rows == 0 || fp.n_frames == 0) return hipSuccess;
    // the stack never holds more than `depth` entries (one per level below the root, plus the root itself)
    if (bvh_depth <= 8)  { if (kernel_name) *kernel_name = "pixel_walk<stack8>";  return launch_stack<8>(sc, fp, count, stream); }
    if (bvh_depth <= 16) { if (kernel_name) *kernel_name = "pixel_walk<stack16>"; return launch_stack<16>(sc, fp, count, stream); }
    if (bvh_depth <= 32) { if (kernel_name) *kernel_name = "pixel_walk<stack32>"; return launch_stack<32>(sc, fp, count, stream); }
    if (bvh_depth <= 64) { if (kernel_name) *kernel_name = "pixel_walk<stack64>"; return launch_stack<64>(sc, fp, count, stream); }
    return hipErrorInvalidValue;     // the reference's own stack is 64 deep (BVHTraversal.cuh:17)
}
#else
bool pixel_walk_built_in() { return false; }
hipError_t launch_render(const SceneView &, const FrameParams &, int, bool, hipStream_t, const char **) { return hipErrorNotSupported; }
#endif

hipError_t launch_kat(int which, const void *d_in, void *d_out, uint32_t n, const FrameParams &fp, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(kat_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, which, (const uint32_t *)d_in, (uint32_t *)d_out, n, fp);
    return hipGetLastError();
}

hipError_t launch_assemble(const void *gathered, void *image, uint32_t width, uint32_t height, uint32_t stripe_rows,
                           uint32_t world, uint32_t padded_rows, hipStream_t stream) {
    if (width == 0 || height == 0) return hipSuccess;
    dim3 grid((width + 255) / 256, height), block(256);
    hipLaunchKernelGGL(assemble_shards_kernel, grid, block, 0, stream, (const float4 *)gathered, (float4 *)image, width,
                       height, stripe_rows, world, padded_rows);
    return hipGetLastError();
}

}  // namespace drt
