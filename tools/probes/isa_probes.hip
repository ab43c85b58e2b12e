// isa_probes.hip -- one kernel per reference operation, calling the SAME device_math.hpp / device_access.hpp function the tracing
// kernel inlines, compiled by tools/isa_by_phase.py with the Makefile's flags.  The VALU issue slots of a probe minus those of
// the empty probe are the cost of ONE call of that operation in the shipped arithmetic (no queueing, no state traffic):
// bench.py's `algorithmic_frac` multiplies them by the exact work counters.  Not part of the library; never run.
#include <hip/hip_runtime.h>

#include "../../dustraytracer_amd/csrc/device_access.hpp"
#include "../../dustraytracer_amd/csrc/device_math.hpp"
#include "../../dustraytracer_amd/csrc/device_scene.hpp"

using namespace drt;

// inputs come from memory and every result goes back: nothing can be folded away
#define LOADF(k) in[(size_t)i * 32 + (k)]
__global__ void probe_empty(const float *in, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    out[i] = LOADF(0);
}
__global__ void probe_tri_test(const float *in, float *out) {                     // Intersection.cu:4-36 + the strict `t < closest` update
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Ray ray; ray.orig = mk3(LOADF(0), LOADF(1), LOADF(2)); ray.dir = mk3(LOADF(3), LOADF(4), LOADF(5)); ray.inv_dir = ray.dir;
    float t, u, v, hit_t = LOADF(15);
    const bool h = tri_intersect_flat(ray, mk3(LOADF(6), LOADF(7), LOADF(8)), mk3(LOADF(9), LOADF(10), LOADF(11)), mk3(LOADF(12), LOADF(13), LOADF(14)), t, u, v);
    int prim = 0;
    if (h && t < hit_t) { hit_t = t; prim = 1; }
    out[i] = hit_t + (float)prim;
}
__global__ void probe_box_test(const float *in, float *out) {                     // Bounds.cu:18-41
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Ray ray; ray.orig = mk3(LOADF(0), LOADF(1), LOADF(2)); ray.dir = ray.orig; ray.inv_dir = mk3(LOADF(3), LOADF(4), LOADF(5));
    out[i] = slab_entry_or_inf(mk3(LOADF(6), LOADF(7), LOADF(8)), mk3(LOADF(9), LOADF(10), LOADF(11)), ray);
}
__global__ void probe_sampler_try(const float *in, float *out) {                  // Random.cu:50-58, one candidate
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t seed = __float_as_uint(LOADF(0));
    f3 p;
    const bool ok = random_unit_sphere_try(seed, p);
    out[i] = p.x + p.y + p.z + (ok ? 1.0f : 0.0f) + __uint_as_float(seed);
}
__global__ void probe_shade_hit(const float *in, float *out) {                    // ClosestHit.cuh:13-24 + RayGen.cuh:121
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Ray ray; ray.orig = mk3(LOADF(0), LOADF(1), LOADF(2)); ray.dir = mk3(LOADF(3), LOADF(4), LOADF(5)); ray.inv_dir = ray.dir;
    f3 position, normal;
    const bool front = closest_hit_frame(ray, LOADF(6), mk3(LOADF(7), LOADF(8), LOADF(9)), position, normal);
    const f3 origin = position + (normal * 0.001f);
    const f3 thr = mk3(LOADF(10), LOADF(11), LOADF(12)) * mk3(LOADF(13), LOADF(14), LOADF(15));       // throughput *= albedo (RayGen.cuh:112)
    out[i] = origin.x + origin.y + origin.z + thr.x + thr.y + thr.z + (front ? 1.0f : 0.0f);
}
__global__ void probe_texture_fetch(const float *in, float *out, SceneView sc, TexDev tex) {      // Texture.cu:33-58 + RayGen.cuh:116
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    TriCold cold;
    for (int k = 0; k < 3; k++) { cold.uv[k][0] = LOADF(2 * k); cold.uv[k][1] = LOADF(2 * k + 1); }
    const f3 c = tex_get_pixel(sc, tex, interp_uv(cold, mk3(LOADF(6), LOADF(7), LOADF(8))));
    out[i] = c.x + c.y + c.z;
}
__global__ void probe_ray_setup(const float *in, float *out) {                    // Ray.cuh:7-9 + the root's slab test (BVHTraversal.cuh:22-26)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const Ray ray = make_ray(mk3(LOADF(0), LOADF(1), LOADF(2)), mk3(LOADF(3), LOADF(4), LOADF(5)) + mk3(LOADF(6), LOADF(7), LOADF(8)));      // :134 N + fuzz
    out[i] = slab_intersect(mk3(LOADF(9), LOADF(10), LOADF(11)), mk3(LOADF(12), LOADF(13), LOADF(14)), ray);
}
__global__ void probe_shadow_ray_setup(const float *in, float *out) {             // RayGen.cuh:124-125: the sun's shadow ray + RayTest's root test
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t seed = __float_as_uint(LOADF(0));
    const Ray ray = make_ray(mk3(LOADF(1), LOADF(2), LOADF(3)), mk3(LOADF(4), LOADF(5), LOADF(6)) + random_unit_vec3(seed) * 1.5f);
    out[i] = slab_intersect(mk3(LOADF(9), LOADF(10), LOADF(11)), mk3(LOADF(12), LOADF(13), LOADF(14)), ray) + __uint_as_float(seed);
}
__global__ void probe_sample(const float *in, float *out, FrameParams fp) {       // per sample: Camera::GetRay, sky, tone map, gamma, seed
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    f2 uv; uv.x = ((float)(i & 1023) / (float)fp.width) * 2 - 1; uv.y = ((float)(i >> 10) / (float)fp.height) * 2 - 1;   // RayGen.cuh:65-66
    uint32_t seed = (uint32_t)i * fp.frame_first;                                                                       // :74-75
    const Ray ray = camera_get_ray(fp, uv, seed);
    f3 light = sky_model(ray.dir, ld3(fp.sky_color)) * mk3(LOADF(0), LOADF(1), LOADF(2)) * fp.sky_intensity;            // :99-108
    light = uncharted2_filmic(light, fp.exposure);                                                                      // :165-169
    light = gamma_correction(light);
    out[i] = light.x + light.y + light.z + ray.orig.x + __uint_as_float(seed);
}
