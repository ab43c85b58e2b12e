// device_math.hpp -- fp32 building blocks of the hot path, device side (gfx950).
//
// Every function states the reference lines it reproduces.  Bit-level rules (DESIGN.md "Numerics"):
//   * compiled with -ffp-contract=off: each * and + is one IEEE rounding, in the reference's order
//   * / and sqrtf are the correctly rounded forms (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt)
//   * rsqrtf(x) is 1.0f / sqrtf(x) (helper_math.cuh:78-81), powf(x, 2) is x * x
//   * fminf/fmaxf lower to v_min_f32/v_max_f32 in IEEE mode: a NaN operand is dropped, as on the CUDA device
// Citations are relative to /root/reference/DustRayTracer/src/.
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>

namespace drt {

struct f3 { float x, y, z; };
struct f2 { float x, y; };

#define DRT_DEV __device__ __forceinline__

DRT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
DRT_DEV f3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }
DRT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
DRT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
DRT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
DRT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
DRT_DEV f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
DRT_DEV f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
DRT_DEV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
DRT_DEV f3 add_scalar(f3 a, float s) { return mk3(a.x + s, a.y + s, a.z + s); }
DRT_DEV f3 sub_scalar(f3 a, float s) { return mk3(a.x - s, a.y - s, a.z - s); }
DRT_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                       // helper_math.cuh:1264
DRT_DEV f3 cross(f3 a, f3 b) {                                                                    // helper_math.cuh:1436
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// ---- exact fast forms of 1/x and sqrt(x) ----
// hipcc's correctly rounded fp32 division / sqrt expansions cost ~42 / ~53 SIMD cycles per wave instruction on gfx950
// (tools/microbench/valu_issue.hip) because they wrap the refinement in v_div_scale / v_div_fmas / v_div_fixup (or range
// scaling and special-case selects) that only matter at the ends of the exponent range.  The two functions below keep the
// refinement and drop the wrapping inside 2^-100 <= |x| <= 2^100, and fall back to the plain operator outside (0, inf, NaN,
// tiny, huge).  They are NOT approximations: tests/test_gpu_parity.py compares each with the plain operator for EVERY one of
// the 2^32 float bit patterns on the device (drt_debug_check_rcp / _sqrt, 0 mismatches required).
DRT_DEV float exact_rcp(float x) {
    // v_rcp_f32 is within 1 ulp; one residual correction with exact FMAs lands on the correctly rounded quotient for every
    // float in the guarded range on gfx950 (tools/microbench/rcp_variants.hip tries the shorter and longer sequences, all 2^32
    // inputs each; the compiler's own expansion spends six FMAs plus the range scaling)
    const float r = __builtin_amdgcn_rcpf(x);
    const float rem = __builtin_fmaf(-x, r, 1.0f);
    float q = __builtin_fmaf(rem, r, r);
    const float ax = __builtin_fabsf(x);
    if (__builtin_expect(!(ax >= 0x1p-100f && ax <= 0x1p100f), 0)) q = 1.0f / x;
    return q;
}
// For a divisor whose small values are thrown away by the caller anyway (the triangle test ignores the result when
// |det| < 1e-6): only the upper end needs the plain operator.  Below 2^-100 the value returned is unspecified.
DRT_DEV float exact_rcp_not_tiny(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    const float rem = __builtin_fmaf(-x, r, 1.0f);
    float q = __builtin_fmaf(rem, r, r);
    if (__builtin_expect(!(__builtin_fabsf(x) <= 0x1p100f), 0)) q = 1.0f / x;
    return q;
}
DRT_DEV float exact_sqrt(float x) {
    if (!(x >= 0x1p-100f && x <= 0x1p100f)) return sqrtf(x);
    // v_sqrt_f32 is within 1 ulp; pick among s-1ulp, s, s+1ulp with exact FMA residuals (the selection step of the
    // compiler's own expansion)
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    float y = (r_dn <= 0.0f) ? s_dn : s;
    y = (r_up > 0.0f) ? s_up : y;
    return y;
}

DRT_DEV float length(f3 v) { return sqrtf(dot(v, v)); }                                           // helper_math.cuh:1307
DRT_DEV f3 normalize(f3 v) { float inv_len = exact_rcp(exact_sqrt(dot(v, v))); return v * inv_len; }   // helper_math.cuh:1325 (1/sqrtf, exact)

// ---- CudaMath/Random.cu ----
DRT_DEV uint32_t pcg_hash(uint32_t input) {                                                       // :6-11
    uint32_t state = input * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
DRT_DEV float random_float(uint32_t &seed) {                                                      // :13-17
    seed = pcg_hash(seed);
    return (float)seed / 4294967296.0f;       // (float)UINT32_MAX == 2^32
}
DRT_DEV f3 random_unit_vec3(uint32_t &seed) {                                                     // :42-48 (x, then y, then z)
    float x = random_float(seed) * 2.f - 1.f;
    float y = random_float(seed) * 2.f - 1.f;
    float z = random_float(seed) * 2.f - 1.f;
    return normalize(mk3(x, y, z));
}
// RNG cycle guard: pcg_hash is a permutation of 2^32 with short cycles (lengths 4, 8, 10, 13, 19, 21, 32, ...); a
// seed that lands on one (RayGen.cuh:91 `seed += i` jumps between cycles) can make every candidate of the two
// rejection loops fail, and the reference then spins forever.  The loops here stop after kMaxTries candidates and
// keep the last one (same rule in oracle/drt_oracle.c); 1024 natural rejections in a row have probability < 1e-180.
constexpr int kMaxTries = 1024;
DRT_DEV f3 random_unit_sphere_vec3(uint32_t &seed) {                                              // :50-58
    for (int tries = 1;; tries++) {
        f3 p = random_unit_vec3(seed);
        float len = length(p);
        if ((len * len) < 1 || tries >= kMaxTries) return p;
    }
}
// One trip of randomUnitSphereVec3's loop (Random.cu:50-58): candidate p and whether the loop returns it.
// The reference accepts when  len*len < 1  with len = sqrtf(dot(p,p)).  For every non-negative float d (and
// inf, NaN)  fl(fl(sqrt(d))^2) < 1  <=>  d < 1  (checked exhaustively over all 2^31 values,
// tests/test_oracle_kat.py::test_sphere_accept_shortcut), so the sqrt and the square are not computed.
// (float)seed / 2^32 * 2 - 1 is written as (float)seed * 2^-31 - 1: both scalings are exact.
DRT_DEV bool random_unit_sphere_try(uint32_t &seed, f3 &p) {
    seed = pcg_hash(seed); float x = (float)seed * 0x1p-31f - 1.f;
    seed = pcg_hash(seed); float y = (float)seed * 0x1p-31f - 1.f;
    seed = pcg_hash(seed); float z = (float)seed * 0x1p-31f - 1.f;
    p = normalize(mk3(x, y, z));
    return dot(p, p) < 1.0f;
}
// randomUnitSphereVec3 (Random.cu:50-58) as a loop over random_unit_sphere_try, with the cycle guard
DRT_DEV f3 random_unit_sphere_vec3_try(uint32_t &seed) {
    f3 p;
    for (int tries = 1;; tries++)
        if (random_unit_sphere_try(seed, p) || tries >= kMaxTries) return p;
}
DRT_DEV f2 random_in_unit_disk(uint32_t &seed) {                                                  // :60-66
    for (int tries = 1;; tries++) {
        f2 p;
        p.x = random_float(seed) * 2 - 1;
        p.y = random_float(seed) * 2 - 1;
        if (p.x * p.x + p.y * p.y < 1.0f || tries >= kMaxTries) return p;
    }
}

// ---- Core/Ray.cuh:5-24 ----
struct Ray { f3 orig, dir, inv_dir; };
DRT_DEV Ray make_ray(f3 o, f3 d) { Ray r; r.orig = o; r.dir = d; r.inv_dir = mk3(exact_rcp(d.x), exact_rcp(d.y), exact_rcp(d.z)); return r; }

// ---- Core/Bounds.cu:18-41 ----
DRT_DEV float slab_intersect(f3 bmin, f3 bmax, const Ray &ray) {
    f3 t0 = (bmin - ray.orig) * ray.inv_dir;
    f3 t1 = (bmax - ray.orig) * ray.inv_dir;
    f3 tmin = mk3(fminf(t0.x, t1.x), fminf(t0.y, t1.y), fminf(t0.z, t1.z));
    f3 tmax = mk3(fmaxf(t1.x, t0.x), fmaxf(t1.y, t0.y), fmaxf(t1.z, t0.z));
    float tenter = fmaxf(fmaxf(tmin.x, tmin.y), tmin.z);
    float texit = fminf(fminf(tmax.x, tmax.y), tmax.z);
    if (tenter < 0.0f) tenter = 0.0f;
    if (tenter > texit || texit < 0) return -1;
    return tenter;
}

// The same test for the closest-hit traversal, returned as "entry distance if the box is hit, +infinity if not".  Both
// uses BVHTraversal.cuh:63-70 makes of the value survive that: `d >= 0 && d < closest` becomes `d < closest` (a hit's d
// is >= 0 or NaN, and NaN fails either form; closest is finite), and the far-child-first order `d1 > d2` only matters
// when both boxes are hit.  Two compare-to-mask instructions (4 cycles each on gfx950) fewer per box.
DRT_DEV float slab_entry_or_inf(f3 bmin, f3 bmax, const Ray &ray) {
    f3 t0 = (bmin - ray.orig) * ray.inv_dir;
    f3 t1 = (bmax - ray.orig) * ray.inv_dir;
    f3 tmin = mk3(fminf(t0.x, t1.x), fminf(t0.y, t1.y), fminf(t0.z, t1.z));
    f3 tmax = mk3(fmaxf(t1.x, t0.x), fmaxf(t1.y, t0.y), fmaxf(t1.z, t0.z));
    float tenter = fmaxf(fmaxf(tmin.x, tmin.y), tmin.z);
    float texit = fminf(fminf(tmax.x, tmax.y), tmax.z);
    if (tenter < 0.0f) tenter = 0.0f;
    return (tenter > texit || texit < 0) ? __builtin_inff() : tenter;
}

// ---- Kernel/Shaders/ClosestHit.cuh:13-24: hit position, and the face normal turned against the ray ----
DRT_DEV bool closest_hit_frame(const Ray &ray, float t, f3 face_n, f3 &position, f3 &normal) {
    position = ray.orig + ray.dir * t;                                       // :13
    const bool back = dot(face_n, normalize(ray.dir)) > 0.f;                 // :17
    normal = back ? (-1.f * face_n) : face_n;
    return !back;                                                            // front_face
}

// ---- Kernel/Shaders/Intersection.cu:4-36 (edges precomputed: e1 = v1-v0, e2 = v2-v0) ----
#define DRT_TRIANGLE_EPSILON 0.000001f     // Common/physical_units.hpp:12
DRT_DEV bool tri_intersect(const Ray &ray, f3 v0, f3 e1, f3 e2, float &t_out, f3 &uvw) {
    f3 pvec = cross(ray.dir, e2);
    float det = dot(e1, pvec);
    if (det > -DRT_TRIANGLE_EPSILON && det < DRT_TRIANGLE_EPSILON) return false;
    float inv_det = 1.0f / det;
    f3 tvec = ray.orig - v0;
    float u = inv_det * dot(tvec, pvec);
    if (u < 0.0f || u > 1.0f) return false;
    f3 qvec = cross(tvec, e1);
    float v = inv_det * dot(ray.dir, qvec);
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = inv_det * dot(e2, qvec);
    if (t > DRT_TRIANGLE_EPSILON) {
        t_out = t;
        uvw = mk3(1.0f - u - v, u, v);
        return true;
    }
    return false;
}

// Same test, straight-line: every quantity is computed, the four rejections are combined at the end.
// A rejected lane may have divided by a tiny or zero det; its u, v, t are then garbage and never used.
// NaN behaves as in the branchy form: comparisons with NaN are false, so only `t > eps` rejects it.
DRT_DEV bool tri_intersect_flat(const Ray &ray, f3 v0, f3 e1, f3 e2, float &t, float &u, float &v) {
    f3 pvec = cross(ray.dir, e2);
    float det = dot(e1, pvec);
    // det > -eps && det < eps  <=>  |det| < eps (NaN: false either way): one compare, the absolute value is an operand modifier
    int ok = !(__builtin_fabsf(det) < DRT_TRIANGLE_EPSILON);
    float inv_det = exact_rcp_not_tiny(det);          // !ok covers |det| < 1e-6: whatever comes back there is not used
    f3 tvec = ray.orig - v0;
    u = inv_det * dot(tvec, pvec);
    f3 qvec = cross(tvec, e1);
    v = inv_det * dot(ray.dir, qvec);
    // (u < 0 | v < 0) and (u > 1 | u + v > 1) through v_min / v_max, which drop a NaN operand exactly as the two separate
    // comparisons ignore it (a compare + mask costs ~4 cycles on gfx950, min/max 2)
    ok = ok & (int)(!(fminf(u, v) < 0.0f)) & (int)(!(fmaxf(u, u + v) > 1.0f));
    t = inv_det * dot(e2, qvec);
    ok = ok & (int)(t > DRT_TRIANGLE_EPSILON);
    return ok != 0;
}

// ---- Shaders/RayGen.cuh:23-61 ----
DRT_DEV f3 uncharted2_tonemap_partial(f3 x) {                                                     // :23-32
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    f3 num = add_scalar(x * add_scalar(A * x, C * B), D * E);
    f3 den = add_scalar(x * add_scalar(A * x, B), D * F);
    return sub_scalar(num / den, E / F);
}
DRT_DEV f3 uncharted2_filmic(f3 v, float exposure) {                                              // :34-42
    f3 curr = uncharted2_tonemap_partial(v * exposure);
    f3 white_scale = mk3(1.0f, 1.0f, 1.0f) / uncharted2_tonemap_partial(mk3(11.2f, 11.2f, 11.2f));
    return curr * white_scale;
}
DRT_DEV f3 gamma_correction(f3 c) { return mk3(exact_sqrt(c.x), exact_sqrt(c.y), exact_sqrt(c.z)); }   // :49-52
DRT_DEV f3 sky_model(f3 dir, f3 sky_color) {                                                      // :54-61
    float t = 0.5f * (1 + normalize(dir).y);      // 0.5 * float in double, rounded back: exact, == 0.5f * float
    f3 c = ((1 - t) * mk3(1, 1, 1)) + (t * sky_color);
    return mk3(c.x * c.x, c.y * c.y, c.z * c.z);
}

}  // namespace drt
