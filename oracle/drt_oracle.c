/*
 * drt_oracle.c -- CPU restatement of DustRayTracer's megakernel hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see drt_oracle.h for the rules and the parity
 * status: leaf arithmetic pinned against oracle/_ref/ref_kat, composition
 * "parity unpinned").
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (never -march=native /
 * -mfma: every * and + below must round separately, exactly as the HIP kernel
 * compiled with -ffp-contract=off does).
 *
 * Decisions the language leaves open in the reference, fixed here (and in the
 * HIP kernel) so that results are reproducible:
 *  - make_float3(randomFloat(s), randomFloat(s), randomFloat(s)) in
 *    CudaMath/Random.cu:44-47 and make_float2(...) in :62 have unspecified
 *    argument evaluation order; we draw x first, then y, then z (what an
 *    LLVM-based device compiler emits; oracle/_ref is built with clang++ for
 *    the same reason).
 *  - rsqrtf(x) := 1.0f / sqrtf(x)  (helper_math.cuh:78-81 host definition).
 *  - std::powf(x, 2) := x * x      (RayGen.cuh:59, Texture.cu:56).
 *  - tan/sin/cos of per-frame constants use the host libm (tanf/sinf/cosf),
 *    hoisted out of the per-pixel code (Camera.cu:85,101; RayGen.cuh:68-71).
 *  - RNG cycle guard.  pcg_hash is a permutation of 2^32 with short cycles (lengths 4, 8, 10, 13, 19, ...;
 *    found exhaustively on the GPU, drt_debug_hash_cycles).  `seed += i` (RayGen.cuh:91) can land on one,
 *    and on some of them every candidate of randomUnitSphereVec3 / random_in_unit_disk is rejected: the
 *    reference then never returns (at 4K x 64 spp x depth 16 this is expected to happen about twice per
 *    image).  Both loops stop after DRT_MAX_TRIES candidates and return the last one; a natural run of that
 *    many rejections has probability < 1e-180, so no terminating reference path is changed.
 *  - Texture::getPixel can index one texel row past the image when
 *    frac(uv) rounds to 1.0 (Texture.cu:35-36): textures carry (width+1)
 *    texels of zero padding so the read is defined.
 *
 * All citations are relative to /root/reference/DustRayTracer/src/.
 */
#include "drt_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } f3;
typedef struct { float x, y; } f2;

/* Optional per-lane step trace (single-threaded runs only; used by tools/sim_schedule.py to study
 * wave scheduling policies): 'N' = stack pop, 'T' = triangle test, 'R' = one try of the rejection
 * sampler, 'S' = end of ray (shade), 'P' = end of path, 'X' = end of pixel. */
static unsigned char *g_trace = NULL;
static size_t g_trace_len = 0, g_trace_cap = 0;
static inline void trace(unsigned char c) { if (g_trace && g_trace_len < g_trace_cap) g_trace[g_trace_len++] = c; }
static __thread int g_trace_passed_u = 0;      /* last tri_intersect got past the u test ('T' in the trace, else 't') */

/* ---- helper_math.cuh subset (each op rounds once; no contraction) ---- */
static inline f3 v3(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f3 add3(f3 a, f3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }          /* :351 */
static inline f3 sub3(f3 a, f3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }          /* :595 */
static inline f3 mul3(f3 a, f3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }          /* :820 */
static inline f3 scale3(f3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }           /* :830 */
static inline f3 scale3l(float s, f3 a) { return v3(s * a.x, s * a.y, s * a.z); }          /* :834 */
static inline f3 adds3(f3 a, float s) { return v3(a.x + s, a.y + s, a.z + s); }            /* :379 family */
static inline f3 subs3(f3 a, float s) { return v3(a.x - s, a.y - s, a.z - s); }            /* :605 */
static inline f3 div3(f3 a, f3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }          /* :1003 */
static inline f3 divs3(f3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }            /* :1013 */
static inline f3 rdiv3(float s, f3 a) { return v3(s / a.x, s / a.y, s / a.z); }            /* :1023 */
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }         /* :1264 */
static inline f3 cross3(f3 a, f3 b) {                                                      /* :1436 */
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float length3(f3 v) { return sqrtf(dot3(v, v)); }                            /* :1307 */
static inline f3 normalize3(f3 v) {                                                        /* :1325 + :78-81 */
    float inv_len = 1.0f / sqrtf(dot3(v, v));
    return scale3(v, inv_len);
}
static inline f3 ld3(const float *p) { return v3(p[0], p[1], p[2]); }
static inline void st3(float *p, f3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

/* ---- CudaMath/Random.cu ---- */
uint32_t o_pcg_hash(uint32_t input)                                                         /* :6-11 */
{
    uint32_t state = input * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

static inline float random_float(uint32_t *seed)                                            /* :13-17 */
{
    *seed = o_pcg_hash(*seed);
    /* (float)UINT32_MAX rounds to 2^32 */
    return (float)*seed / 4294967296.0f;
}

static inline f3 random_unit_vec3(uint32_t *seed)                                           /* :42-48 */
{
    float x = random_float(seed) * 2.f - 1.f;
    float y = random_float(seed) * 2.f - 1.f;
    float z = random_float(seed) * 2.f - 1.f;
    return normalize3(v3(x, y, z));
}

#define DRT_MAX_TRIES 1024                   /* RNG cycle guard, see the header comment */

static inline f3 random_unit_sphere_vec3(uint32_t *seed, uint64_t *iters)                   /* :50-58 */
{
    for (int tries = 1;; tries++) {
        f3 p = random_unit_vec3(seed);
        float len = length3(p);
        if (iters) ++*iters;
        trace('R');
        if ((len * len) < 1 || tries >= DRT_MAX_TRIES)
            return p;
    }
}

static inline f2 random_in_unit_disk(uint32_t *seed)                                        /* :60-66 */
{
    for (int tries = 1;; tries++) {
        f2 p;
        p.x = random_float(seed) * 2 - 1;
        p.y = random_float(seed) * 2 - 1;
        if (p.x * p.x + p.y * p.y < 1.0f || tries >= DRT_MAX_TRIES)
            return p;
    }
}

/* ---- Core/Ray.cuh:5-24 ---- */
typedef struct { f3 orig, dir, inv_dir; } ray_t;

static inline ray_t make_ray(f3 o, f3 d)
{
    ray_t r; r.orig = o; r.dir = d; r.inv_dir = rdiv3(1.0f, d); return r;
}

/* ---- Core/Bounds.cu:18-41 ---- */
static inline float slab_intersect(const float *bmin, const float *bmax, const ray_t *ray)
{
    f3 t0 = mul3(sub3(ld3(bmin), ray->orig), ray->inv_dir);
    f3 t1 = mul3(sub3(ld3(bmax), ray->orig), ray->inv_dir);
    /* device fminf/fmaxf: IEEE minNum/maxNum (a NaN operand is dropped) */
    f3 tmin = v3(fminf(t0.x, t1.x), fminf(t0.y, t1.y), fminf(t0.z, t1.z));
    f3 tmax = v3(fmaxf(t1.x, t0.x), fmaxf(t1.y, t0.y), fmaxf(t1.z, t0.z));
    float tenter = fmaxf(fmaxf(tmin.x, tmin.y), tmin.z);
    float texit = fminf(fminf(tmax.x, tmax.y), tmax.z);
    if (tenter < 0.0f)
        tenter = 0.0f;
    if (tenter > texit || texit < 0)
        return -1;
    return tenter;
}

/* ---- Kernel/Shaders/Intersection.cu:4-36 ---- */
#define TRIANGLE_EPSILON 0.000001f       /* Common/physical_units.hpp:12 */

typedef struct { float t; f3 uvw; int hit; } short_hit;

static inline short_hit tri_intersect(const ray_t *ray, const float p[3][3])
{
    short_hit out; out.t = -1; out.hit = 0; out.uvw = v3(0, 0, 0);
    f3 p0 = ld3(p[0]);
    f3 v0v1 = sub3(ld3(p[1]), p0);
    f3 v0v2 = sub3(ld3(p[2]), p0);
    f3 pvec = cross3(ray->dir, v0v2);
    float det = dot3(v0v1, pvec);
    if (det > -TRIANGLE_EPSILON && det < TRIANGLE_EPSILON)
        return out;
    float inv_det = 1.0f / det;
    f3 tvec = sub3(ray->orig, p0);
    float u = inv_det * dot3(tvec, pvec);
    if (u < 0.0f || u > 1.0f)
        return out;
    g_trace_passed_u = 1;
    f3 qvec = cross3(tvec, v0v1);
    float v = inv_det * dot3(ray->dir, qvec);
    if (v < 0.0f || u + v > 1.0f)
        return out;
    float t = inv_det * dot3(v0v2, qvec);
    if (t > TRIANGLE_EPSILON) {
        out.t = t; out.hit = 1;
        out.uvw = v3(1.0f - u - v, u, v);
    }
    return out;
}

/* ---- Core/Scene/Texture.cu:33-75 ---- */
static inline f3 tex_get_pixel(const o_texture *tex, f2 uv)                                 /* :33-58 */
{
    int x = (int)((uv.x - floorf(uv.x)) * tex->width);
    int y = (int)((uv.y - floorf(uv.y)) * tex->height);
    uint8_t r = 0, g = 0, b = 255;
    if (tex->comps == 3) {
        const uint8_t *t = tex->data + 3 * ((size_t)y * tex->width + x);
        r = t[0]; g = t[1]; b = t[2];
    } else if (tex->comps == 4) {
        const uint8_t *t = tex->data + 4 * ((size_t)y * tex->width + x);
        r = t[0]; g = t[1]; b = t[2];
    }
    f3 c = v3(r / (float)255, g / (float)255, b / (float)255);
    return v3(c.x * c.x, c.y * c.y, c.z * c.z);
}

static inline float tex_get_alpha(const o_texture *tex, f2 uv)                              /* :60-75 */
{
    if (tex->comps < 4)
        return 1;
    int x = (int)((uv.x - floorf(uv.x)) * tex->width);
    int y = (int)((uv.y - floorf(uv.y)) * tex->height);
    uint8_t a = tex->data[4 * ((size_t)y * tex->width + x) + 3];
    return a / (float)255;
}

static inline f2 interp_uv(const o_triangle *tri, f3 uvw)            /* RayGen.cuh:116, AnyHit.cuh:20-22 */
{
    f2 r;
    r.x = uvw.x * tri->uv[0][0] + uvw.y * tri->uv[1][0] + uvw.z * tri->uv[2][0];
    r.y = uvw.x * tri->uv[0][1] + uvw.y * tri->uv[1][1] + uvw.z * tri->uv[2][1];
    return r;
}

/* ---- Kernel/Shaders/AnyHit.cuh:8-28 ---- */
static inline int any_hit(const o_scene *sc, const o_triangle *tri, f3 uvw, o_counters *cnt)
{
    const o_material *m = &sc->mats[tri->material];
    if (m->albedo_tex < 0)
        return 1;
    const o_texture *tex = &sc->texs[m->albedo_tex];
    if (tex->comps < 4)
        return 1;
    f2 uv = interp_uv(tri, uvw);
    if (cnt) cnt->anyhit_alpha++;
    float alpha = tex_get_alpha(tex, uv);
    return !(alpha < 1);
}

/* ---- Core/HitPayload.cuh:8-19 ---- */
typedef struct {
    float t;
    f3 normal, position, color, uvw;
    const o_triangle *prim;
    int front_face;                    /* HitPayload.cuh:10 (the reference stores it and never reads it; the opt-in dielectric lobe does) */
} hit_payload;

#define STACK_SIZE 64     /* BVH/BVHTraversal.cuh:17 */

/* Interval::surrounds (Interval.cuh:24-26) */
static int interval_surrounds(float imin, float imax, float x) { return imin < x && x < imax; }
/* Bounds3f::getCentroid (Bounds.cu:12-15) */
static f3 bounds_centroid(f3 pmin, f3 pmax) { return add3(scale3l(0.5f, pmin), scale3l(0.5f, pmax)); }

/* ---- BVH/BVHTraversal.cuh:14-73 ---- */
static void traverse_bvh(const ray_t *ray, int root, hit_payload *closest, const o_scene *sc, o_counters *cnt)
{
    if (root < 0) return;
    int idx_stack[STACK_SIZE];
    float dist_stack[STACK_SIZE];
    int sp = 0;
    const float interval_min = -1.0f, interval_max = FLT_MAX;     /* RayGen.cuh:78 */

    idx_stack[sp] = root;
    dist_stack[sp++] = slab_intersect(sc->nodes[root].bmin, sc->nodes[root].bmax, ray);

    while (sp > 0) {
        const o_bvh_node *node = &sc->nodes[idx_stack[--sp]];
        float node_dist = dist_stack[sp];
        trace('N');
        if (!interval_surrounds(interval_min, interval_max, node_dist)) continue;        /* :38 ray.interval.surrounds(current_node_hitdist) */
        if (closest->prim != NULL && closest->t < node_dist) continue;                   /* :41 */
        closest->color = add3(closest->color, scale3(v3(1, 1, 1), 0.05f));               /* :43 */
        if (cnt) cnt->node_visits++;

        if (node->is_leaf) {
            for (int i = node->prim_start; i < node->prim_start + node->prim_count; i++) {
                const o_triangle *tri = &sc->tris[i];
                g_trace_passed_u = 0;
                short_hit h = tri_intersect(ray, tri->p);
                trace(g_trace_passed_u ? 'T' : 't');
                if (cnt) cnt->tri_tests++;
                if (h.hit && h.t < closest->t) {                                         /* :51 */
                    if (!any_hit(sc, tri, h.uvw, cnt)) continue;
                    closest->t = h.t;
                    closest->prim = tri;
                    closest->uvw = h.uvw;
                }
            }
        } else {
            if (cnt) cnt->inner_visits++;
            const o_bvh_node *c1 = &sc->nodes[node->child1], *c2 = &sc->nodes[node->child2];
            float d1 = slab_intersect(c1->bmin, c1->bmax, ray);
            float d2 = slab_intersect(c2->bmin, c2->bmax, ray);
            if (d1 > d2) {                                                               /* :63-70 */
                if (d1 >= 0 && d1 < closest->t) { dist_stack[sp] = d1; idx_stack[sp++] = node->child1; }
                if (d2 >= 0 && d2 < closest->t) { dist_stack[sp] = d2; idx_stack[sp++] = node->child2; }
            } else {
                if (d2 >= 0 && d2 < closest->t) { dist_stack[sp] = d2; idx_stack[sp++] = node->child2; }
                if (d1 >= 0 && d1 < closest->t) { dist_stack[sp] = d1; idx_stack[sp++] = node->child1; }
            }
            if (cnt && (uint64_t)sp > cnt->max_stack) cnt->max_stack = (uint64_t)sp;
        }
    }
}

/* ---- BVH/BVHTraversal.cuh:76-134 ---- */
static int traverse_bvh_raytest(const ray_t *ray, int root, const o_scene *sc, o_counters *cnt)
{
    if (root < 0) return 0;
    int idx_stack[STACK_SIZE];
    int sp = 0;
    idx_stack[sp++] = root;

    while (sp > 0) {
        int idx = idx_stack[--sp];
        const o_bvh_node *node = &sc->nodes[idx];
        if (idx == root) {                                                               /* :95-103 */
            float d = slab_intersect(node->bmin, node->bmax, ray);
            if (d < 0) continue;
        }
        if (node->is_leaf) {
            for (int i = node->prim_start; i < node->prim_start + node->prim_count; i++) {
                const o_triangle *tri = &sc->tris[i];
                short_hit h = tri_intersect(ray, tri->p);
                if (cnt) cnt->tri_tests_shadow++;
                if (h.hit && any_hit(sc, tri, h.uvw, cnt))
                    return 1;
            }
        } else {
            if (cnt) cnt->inner_visits_shadow++;
            const o_bvh_node *c1 = &sc->nodes[node->child1], *c2 = &sc->nodes[node->child2];
            float h1 = slab_intersect(c1->bmin, c1->bmax, ray);
            float h2 = slab_intersect(c2->bmin, c2->bmax, ray);
            if (h1 > h2) {                                                               /* :122-129 */
                if (h1 >= 0) idx_stack[sp++] = node->child1;
                if (h2 >= 0) idx_stack[sp++] = node->child2;
            } else {
                if (h2 >= 0) idx_stack[sp++] = node->child2;
                if (h1 >= 0) idx_stack[sp++] = node->child1;
            }
            if (cnt && (uint64_t)sp > cnt->max_stack) cnt->max_stack = (uint64_t)sp;
        }
    }
    return 0;
}

/* Shaders/ClosestHit.cuh:13-24: hit position, and the face normal turned against the ray.  Returns front_face. */
static int closest_hit_frame(const ray_t *ray, float t, f3 fn, f3 *position, f3 *normal)
{
    *position = add3(ray->orig, scale3(ray->dir, t));                                       /* :13 */
    if (dot3(fn, normalize3(ray->dir)) > 0.f) {                                             /* :17 */
        *normal = scale3l(-1.f, fn);
        return 0;
    }
    *normal = fn;
    return 1;
}

/* ---- Kernel/TraceRay.cu:15-32, Shaders/ClosestHit.cuh:4-28, Shaders/Miss.cuh:2-6 ---- */
static hit_payload trace_ray(const ray_t *ray, const o_scene *sc, o_counters *cnt)
{
    hit_payload w;
    memset(&w, 0, sizeof w);
    w.prim = NULL;
    w.t = FLT_MAX;                                       /* ray.interval.max, RayGen.cuh:78 */
    w.color = v3(0, 0, 0);
    if (cnt) cnt->rays++;
    traverse_bvh(ray, sc->n_nodes - 1, &w, sc, cnt);

    hit_payload out;
    memset(&out, 0, sizeof out);
    out.color = w.color;
    if (w.prim == NULL) {                                /* Miss */
        out.prim = NULL;
        out.t = -1;
        return out;
    }
    out.prim = w.prim;                                   /* ClosestHit */
    out.uvw = w.uvw;
    out.t = w.t;
    out.front_face = closest_hit_frame(ray, w.t, ld3(w.prim->face_n), &out.position, &out.normal);
    return out;
}

/* ---- Shaders/RayGen.cuh:23-61 ---- */
static inline f3 uncharted2_tonemap_partial(f3 x)                                           /* :23-32 */
{
    float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    f3 num = adds3(mul3(x, adds3(scale3l(A, x), C * B)), D * E);
    f3 den = adds3(mul3(x, adds3(scale3l(A, x), B)), D * F);
    return subs3(div3(num, den), E / F);
}

static inline f3 uncharted2_filmic(f3 v, float exposure)                                    /* :34-42 */
{
    f3 curr = uncharted2_tonemap_partial(scale3(v, exposure));
    f3 W = v3(11.2f, 11.2f, 11.2f);
    f3 white_scale = div3(v3(1.0f, 1.0f, 1.0f), uncharted2_tonemap_partial(W));
    return mul3(curr, white_scale);
}

static inline f3 gamma_correction(f3 c) { return v3(sqrtf(c.x), sqrtf(c.y), sqrtf(c.z)); }  /* :49-52 */

static inline f3 sky_model(const ray_t *ray, const o_settings *s)                           /* :54-61 */
{
    /* 0.5 * (1 + n.y) is a double product of a float sum: exact, same as 0.5f*(...) */
    float t = (float)(0.5 * (double)(1 + normalize3(ray->dir).y));
    f3 col1 = ld3(s->sky_color);
    f3 col2 = v3(1, 1, 1);
    f3 c = add3(scale3l((float)(1 - t), col2), scale3l(t, col1));
    return v3(c.x * c.x, c.y * c.y, c.z * c.z);
}

/* Per-frame constants of Camera::GetRay (Camera.cu:84-103), hoisted. */
typedef struct {
    f3 position, fwd_focus, horizontal, vertical, disk_u, disk_v;
    int defocus;
    float exposure;
} cam_frame;

static cam_frame camera_frame(const o_camera *cam, float width, float height)
{
    cam_frame cf;
    float theta = cam->vfov_rad / 2;
    float fov_factor = tanf(theta / 2.0f);                                                  /* :84-85 */
    float aspect_ratio = width / height;
    float plane_h = 2.0f * fov_factor * cam->focus_dist;
    float plane_w = plane_h * aspect_ratio;
    f3 forward_dir = normalize3(ld3(cam->forward));
    f3 right_dir = normalize3(cross3(forward_dir, v3(0, 1, 0)));
    f3 up_dir = cross3(right_dir, forward_dir);
    cf.horizontal = scale3l(plane_w, right_dir);
    cf.vertical = scale3l(plane_h, up_dir);
    float PI = 3.14159265359f;                                                              /* :125-129 */
    float defocus_radius = cam->focus_dist * tanf((cam->defocus_angle * (PI / 180.f)) / 2.0f);
    cf.disk_u = scale3l(defocus_radius, right_dir);
    cf.disk_v = scale3l(defocus_radius, up_dir);
    cf.defocus = !(cam->defocus_angle <= 0);
    cf.position = ld3(cam->position);
    cf.fwd_focus = scale3(forward_dir, cam->focus_dist);
    cf.exposure = cam->exposure;
    return cf;
}

static inline ray_t camera_get_ray(const cam_frame *cf, f2 uv, uint32_t *seed)              /* Camera.cu:98-122 */
{
    f2 offset;
    offset.x = random_float(seed) - 0.5f;
    offset.y = random_float(seed) - 0.5f;
    offset.x *= 0.0035f; offset.y *= 0.0035f;
    f3 rorig;
    if (!cf->defocus) {
        rorig = cf->position;
    } else {
        f2 p = random_in_unit_disk(seed);
        rorig = add3(add3(cf->position, scale3l(p.x, cf->disk_u)), scale3l(p.y, cf->disk_v));
    }
    f3 d = add3(sub3(add3(add3(cf->fwd_focus, scale3l(uv.x + offset.x, cf->horizontal)),
                          scale3l(uv.y + offset.y, cf->vertical)), rorig), cf->position);
    return make_ray(rorig, normalize3(d));
}

/* Per-frame constants of RayGen (RayGen.cuh:68-72), hoisted. */
typedef struct { f3 sunpos, suncol; } sun_frame;

static sun_frame sun_frame_of(const o_settings *s)
{
    sun_frame sf;
    float sx = sinf(s->sunlight_dir[0]), sy = sinf(s->sunlight_dir[1]), cx = cosf(s->sunlight_dir[0]);
    sf.sunpos = scale3(v3(sx * (1 - sy), sy, cx * (1 - sy)), 100);
    sf.suncol = scale3(ld3(s->sunlight_color), s->sunlight_intensity);
    return sf;
}

/* ---- Shaders/RayGen.cuh:63-172 ---- */
static f3 ray_gen(uint32_t x, uint32_t y, uint32_t max_x, uint32_t max_y, const cam_frame *cf,
                  const sun_frame *sf, uint32_t frameidx, const o_scene *sc, const o_settings *set,
                  o_counters *cnt)
{
    f2 screen_uv;
    screen_uv.x = ((float)x / (float)max_x) * 2 - 1;                                        /* :65-66 */
    screen_uv.y = ((float)y / (float)max_y) * 2 - 1;

    uint32_t seed = x + y * max_x;                                                          /* :74-75 */
    seed *= frameidx;

    ray_t ray = camera_get_ray(cf, screen_uv, &seed);

    f3 light = v3(0, 0, 0), throughput = v3(1, 1, 1);
    int bounces = set->ray_bounce_limit;
    f2 tex_uv = { 0, 1 };
    const int debug = set->render_mode == 1;
    if (cnt) cnt->samples++;

    for (int i = 0; i <= bounces; i++) {                                                    /* :88 */
        hit_payload payload = trace_ray(&ray, sc, cnt);
        seed += (uint32_t)i;                                                                /* :91 */
        trace('S');

        if (payload.prim == NULL) {                                                         /* :99-108 */
            if (set->debug_mode == 4 && debug) {
                light = payload.color;
            } else {
                f3 sky = sky_model(&ray, set);
                light = add3(light, scale3(mul3(sky, throughput), set->sky_intensity));
            }
            break;
        }

        const o_material *mat = &sc->mats[payload.prim->material];                          /* :111-118 */
        const o_material_ext *ext = sc->mats_ext ? &sc->mats_ext[payload.prim->material] : NULL;
        if (ext && sc->ext_emissive && !debug)                       /* opt-in (o_scene): emission seen through the path so far */
            light = add3(light, mul3(scale3(ld3(ext->emissive), sc->ext_emissive_scale), throughput));
        if (mat->albedo_tex < 0) {
            throughput = mul3(throughput, ld3(mat->albedo));
            if (cnt) cnt->hits_flat++;
        } else {
            tex_uv = interp_uv(payload.prim, payload.uvw);
            throughput = mul3(throughput, tex_get_pixel(&sc->texs[mat->albedo_tex], tex_uv));
            if (cnt) cnt->hits_textured++;
        }

        f3 new_origin = add3(payload.position, scale3(payload.normal, 0.001f));             /* :121 */

        if (set->enable_sunlight && !debug) {                                               /* :124-128 */
            ray_t shadow = make_ray(new_origin, add3(sf->sunpos, scale3(random_unit_vec3(&seed), 1.5f)));
            if (cnt) cnt->shadow_rays++;
            if (!traverse_bvh_raytest(&shadow, sc->n_nodes - 1, sc, cnt))
                light = add3(light, mul3(sf->suncol, throughput));
        }

        const uint64_t iters_before = cnt ? cnt->sphere_iters : 0;
        if (ext && sc->ext_transmission && ext->transmission && !debug) {
            /* opt-in: dielectric interface (the reference's refract / reflectance, CudaMath/Random.cu:26-40, which nothing calls) */
            f3 v = normalize3(ray.dir);
            float cos_theta = fminf(dot3(scale3(v, -1.f), payload.normal), 1.0f);                /* Random.cu:28 */
            float ri = payload.front_face ? 1.0f / ext->refractive_index : ext->refractive_index;
            float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
            int reflect = ri * sin_theta > 1.0f;
            float r0 = (1 - ri) / (1 + ri);                                                       /* Random.cu:37-39 */
            r0 = r0 * r0;
            float one_minus = 1 - cos_theta;
            float p5 = ((one_minus * one_minus) * (one_minus * one_minus)) * one_minus;          /* pow(1 - cosine, 5) */
            float refl = r0 + (1 - r0) * p5;
            if (!reflect) reflect = refl > random_float(&seed);
            if (reflect) {
                f3 dir = sub3(v, scale3(payload.normal, 2.0f * dot3(v, payload.normal)));
                ray = make_ray(new_origin, dir);
            } else {
                f3 perp = scale3(add3(v, scale3(payload.normal, cos_theta)), ri);                 /* Random.cu:29 */
                f3 par = scale3(payload.normal, -sqrtf(fabsf(1.0f - dot3(perp, perp))));          /* :30 */
                ray = make_ray(sub3(payload.position, scale3(payload.normal, 0.001f)), add3(perp, par));
            }
            continue;
        }
        f3 fuzz = random_unit_sphere_vec3(&seed, cnt ? &cnt->sphere_iters : NULL);
        if (cnt && i < bounces) cnt->sphere_iters_traced += cnt->sphere_iters - iters_before;
        if (ext && sc->ext_specular && ext->metallic && !debug) {   /* opt-in: mirror lobe, fuzzed by the roughness */
            f3 v = normalize3(ray.dir);
            f3 refl = sub3(v, scale3(payload.normal, 2.0f * dot3(v, payload.normal)));
            f3 dir = add3(refl, scale3(fuzz, ext->roughness));
            if (!(dot3(dir, payload.normal) > 0.0f)) break;          /* scattered into the surface: absorbed */
            ray = make_ray(new_origin, dir);
        } else
        ray = make_ray(new_origin, add3(payload.normal, fuzz));                             /* :133-134 */

        if (debug) {                                                                        /* :137-161 */
            switch (set->debug_mode) {
            case 0: light = throughput; break;
            case 1: light = payload.normal; break;
            case 2: light = payload.uvw; break;
            case 3: light = v3(tex_uv.x, tex_uv.y, 0); break;
            case 4: light = add3(v3(0, 0.1f, 0.1f), payload.color); break;
            default: break;
            }
            break;
        }
    }

    trace('P');
    if (!debug || set->debug_mode == 0) {                                                   /* :165-169 */
        if (set->tone_mapping) light = uncharted2_filmic(light, cf->exposure);
        if (set->gamma_correction) light = gamma_correction(light);
    }
    return light;
}

/* ---- Kernel/RenderKernel.cu:20-35 over a set of rows ---- */
typedef struct {
    const o_scene *scene; const o_settings *set;
    cam_frame cf; sun_frame sf;
    int32_t W, H; uint32_t frame_first, n_frames;
    float *accum, *rgba;
    int32_t stripe_rows, rank, world;
    volatile int32_t *next_row;
    o_counters counters;
    int want_counters;
} job_t;

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    o_counters *cnt = j->want_counters ? &j->counters : NULL;
    for (;;) {
        int32_t y = __sync_fetch_and_add(j->next_row, 1);
        if (y >= j->H) break;
        if ((y / j->stripe_rows) % j->world != j->rank) continue;
        for (int32_t x = 0; x < j->W; x++) {
            size_t p = (size_t)x + (size_t)y * (size_t)j->W;
            f3 acc = ld3(&j->accum[3 * p]);
            uint32_t f = j->frame_first;
            for (uint32_t k = 0; k < j->n_frames; k++, f++) {
                f3 c = ray_gen((uint32_t)x, (uint32_t)y, (uint32_t)j->W, (uint32_t)j->H,
                               &j->cf, &j->sf, f, j->scene, j->set, cnt);
                acc = add3(acc, c);                                                         /* :29 */
            }
            trace('X');
            st3(&j->accum[3 * p], acc);
            f3 out = divs3(acc, (float)(f - 1));                                            /* :30 */
            j->rgba[4 * p + 0] = out.x; j->rgba[4 * p + 1] = out.y;
            j->rgba[4 * p + 2] = out.z; j->rgba[4 * p + 3] = 1;
        }
    }
    return NULL;
}

static void add_counters(o_counters *a, const o_counters *b)
{
    a->samples += b->samples; a->rays += b->rays; a->node_visits += b->node_visits;
    a->inner_visits += b->inner_visits; a->tri_tests += b->tri_tests;
    a->hits_textured += b->hits_textured; a->hits_flat += b->hits_flat;
    a->shadow_rays += b->shadow_rays; a->inner_visits_shadow += b->inner_visits_shadow;
    a->tri_tests_shadow += b->tri_tests_shadow; a->anyhit_alpha += b->anyhit_alpha;
    a->sphere_iters += b->sphere_iters;
    a->sphere_iters_traced += b->sphere_iters_traced;
    if (b->max_stack > a->max_stack) a->max_stack = b->max_stack;
}

void o_render(const o_scene *scene, const o_camera *cam, const o_settings *set,
              int32_t W, int32_t H, uint32_t frame_first, uint32_t n_frames,
              float *accum, float *rgba, int32_t n_threads,
              int32_t stripe_rows, int32_t rank, int32_t world, o_counters *counters)
{
    if (n_threads <= 0) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    if (stripe_rows <= 0) stripe_rows = 1;
    if (world <= 0) { world = 1; rank = 0; }
    if (n_frames == 0 || W <= 0 || H <= 0) return;

    volatile int32_t next_row = 0;
    job_t *jobs = (job_t *)calloc((size_t)n_threads, sizeof(job_t));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    /* Camera.cu:82: width/height arrive as float */
    cam_frame cf = camera_frame(cam, (float)(uint32_t)W, (float)(uint32_t)H);
    sun_frame sf = sun_frame_of(set);
    for (int t = 0; t < n_threads; t++) {
        job_t *j = &jobs[t];
        j->scene = scene; j->set = set; j->cf = cf; j->sf = sf;
        j->W = W; j->H = H; j->frame_first = frame_first; j->n_frames = n_frames;
        j->accum = accum; j->rgba = rgba;
        j->stripe_rows = stripe_rows; j->rank = rank; j->world = world;
        j->next_row = &next_row; j->want_counters = counters != NULL;
    }
    for (int t = 1; t < n_threads; t++) pthread_create(&th[t], NULL, worker, &jobs[t]);
    worker(&jobs[0]);
    for (int t = 1; t < n_threads; t++) pthread_join(th[t], NULL);
    if (counters)
        for (int t = 0; t < n_threads; t++) add_counters(counters, &jobs[t].counters);
    free(jobs); free(th);
}

/* Single-threaded render of a pixel rectangle that records the step trace (see g_trace). Returns bytes written. */
size_t o_trace_steps(const o_scene *scene, const o_camera *cam, const o_settings *set, int32_t W, int32_t H,
                     int32_t x0, int32_t y0, int32_t x1, int32_t y1, uint32_t n_frames, unsigned char *buf, size_t cap)
{
    cam_frame cf = camera_frame(cam, (float)(uint32_t)W, (float)(uint32_t)H);
    sun_frame sf = sun_frame_of(set);
    g_trace = buf; g_trace_len = 0; g_trace_cap = cap;
    for (int32_t y = y0; y < y1; y++)
        for (int32_t x = x0; x < x1; x++) {
            for (uint32_t f = 1; f <= n_frames; f++)
                (void)ray_gen((uint32_t)x, (uint32_t)y, (uint32_t)W, (uint32_t)H, &cf, &sf, f, scene, set, NULL);
            trace('X');
        }
    g_trace = NULL;
    return g_trace_len;
}

void o_default_settings(o_settings *s)                               /* Scene/RendererSettings.h:22-34 */
{
    memset(s, 0, sizeof *s);
    s->gamma_correction = 1; s->tone_mapping = 1; s->enable_sunlight = 0;
    s->max_samples = 500; s->ray_bounce_limit = 2; s->render_mode = 0; s->debug_mode = 0;
    s->sunlight_dir[0] = -0.803f; s->sunlight_dir[1] = 0.681f;
    s->sunlight_color[0] = 1.000f; s->sunlight_color[1] = 0.944f; s->sunlight_color[2] = 0.917f;
    s->sunlight_intensity = 30;
    s->sky_color[0] = 0.25f; s->sky_color[1] = 0.498f; s->sky_color[2] = 0.80f;
    s->sky_intensity = 20;
}

void o_default_camera(o_camera *c)                 /* Scene/Camera.cuh:32-46, Editor/EditorLayer.cpp:35-40 */
{
    float PI = 3.14159265359f;
    c->exposure = 1;
    c->vfov_rad = 60 * (PI / 180.f);
    c->defocus_angle = 0;
    c->focus_dist = 10;
    c->position[0] = 0; c->position[1] = 2; c->position[2] = 5;
    c->forward[0] = 0; c->forward[1] = 0; c->forward[2] = -1;
}

/* ================= host prep ================= */

/* Scene/Scene.cu:272-302 + Scene/Triangle.cuh:9-12 */
void o_build_triangles(const float *pos, const float *nrm, const float *uv,
                       const int32_t *mat, int32_t n_tris, o_triangle *out)
{
    for (int32_t t = 0; t < n_tris; t++) {
        o_triangle *tri = &out[t];
        f3 P[3], N[3];
        for (int k = 0; k < 3; k++) {
            P[k] = ld3(&pos[(3 * t + k) * 3]);
            N[k] = ld3(&nrm[(3 * t + k) * 3]);
            st3(tri->p[k], P[k]); st3(tri->n[k], N[k]);
            tri->uv[k][0] = uv[(3 * t + k) * 2 + 0];
            tri->uv[k][1] = uv[(3 * t + k) * 2 + 1];
        }
        f3 e0 = sub3(P[1], P[0]), e1 = sub3(P[2], P[0]);                                    /* :275-277 */
        f3 face = cross3(e0, e1);
        f3 avg = divs3(add3(add3(N[0], N[1]), N[2]), 3);                                    /* :279 */
        float ndot = dot3(face, avg);
        f3 sn = (ndot < 0.0f) ? v3(-face.x, -face.y, -face.z) : face;                       /* :282 */
        st3(tri->face_n, normalize3(sn));                                                   /* :300 */
        st3(tri->centroid, divs3(add3(add3(P[0], P[1]), P[2]), 3));                         /* Triangle.cuh:11 */
        tri->material = mat[t];
    }
}

/* BVH/BVHBuilder.cuh:48-95 -- returns extent = max-min, min in *mn */
static f3 absolute_extent(const o_triangle *tris, const int32_t *idx, int32_t start, int32_t end, f3 *mn)
{
    f3 lo = v3(FLT_MAX, FLT_MAX, FLT_MAX), hi = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int32_t i = start; i < end; i++) {
        const o_triangle *t = &tris[idx ? idx[i] : i];
        for (int k = 0; k < 3; k++) {
            lo.x = fminf(lo.x, t->p[k][0]); lo.y = fminf(lo.y, t->p[k][1]); lo.z = fminf(lo.z, t->p[k][2]);
            hi.x = fmaxf(hi.x, t->p[k][0]); hi.y = fmaxf(hi.y, t->p[k][1]); hi.z = fmaxf(hi.z, t->p[k][2]);
        }
    }
    *mn = lo;
    return v3(hi.x - lo.x, hi.y - lo.y, hi.z - lo.z);
}

static float surface_area(const float *bmin, const float *bmax)                             /* Bounds.cu:4-10 */
{
    float planex = 2 * (bmax[2] - bmin[2]) * (bmax[1] - bmin[1]);
    float planey = 2 * (bmax[2] - bmin[2]) * (bmax[0] - bmin[0]);
    float planez = 2 * (bmax[0] - bmin[0]) * (bmax[1] - bmin[1]);
    return planex + planey + planez;
}

static float node_surface_area(const o_bvh_node *n)                                         /* BVHNode.cuh:29-35 */
{
    if (n->prim_count == 0) return 0;
    return surface_area(n->bmin, n->bmax);
}

static void set_bounds(o_bvh_node *n, f3 mn, f3 ext)            /* Bounds3f(min, min + extent) */
{
    st3(n->bmin, mn);
    st3(n->bmax, add3(mn, ext));
}

/* float -> int as x86-64 cvttss2si does (the reference is an x64 build): NaN/overflow -> INT_MIN */
static int32_t f2i_x86(float f)
{
    if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT_MIN;
    return (int32_t)f;
}

static float centroid_axis(const o_triangle *t, int axis) { return t->centroid[axis]; }

/* BVHBuilder.cu:216-255 */
static void bin_to_shallow_nodes(o_bvh_node *left, o_bvh_node *right, float bin, int axis,
                                 const o_triangle *tris, int32_t start, int32_t end,
                                 int32_t *lidx, int32_t *ridx)
{
    int32_t nl = 0, nr = 0;
    for (int32_t i = start; i < end; i++) {
        if (centroid_axis(&tris[i], axis) < bin) lidx[nl++] = i; else ridx[nr++] = i;
    }
    f3 mn, ext;
    left->prim_count = nl;
    ext = absolute_extent(tris, lidx, 0, nl, &mn);
    set_bounds(left, mn, ext);
    right->prim_count = nr;
    ext = absolute_extent(tris, ridx, 0, nr, &mn);
    set_bounds(right, mn, ext);
}

/* libstdc++ std::partition for bidirectional iterators (bits/stl_algo.h __partition):
 * advance first over "true", retreat last over "false", swap, repeat. */
static int32_t partition_tris(o_triangle *tris, int32_t first, int32_t last, float bin, int axis)
{
    for (;;) {
        for (;;) {
            if (first == last) return first;
            else if (centroid_axis(&tris[first], axis) < bin) ++first;
            else break;
        }
        --last;
        for (;;) {
            if (first == last) return first;
            else if (!(centroid_axis(&tris[last], axis) < bin)) --last;
            else break;
        }
        o_triangle tmp = tris[first]; tris[first] = tris[last]; tris[last] = tmp;
        ++first;
    }
}

/* BVHBuilder.cu:175-214 */
static void bin_to_nodes(o_bvh_node *left, o_bvh_node *right, float bin, int axis,
                         o_triangle *tris, int32_t start, int32_t end)
{
    int32_t mid = partition_tris(tris, start, end, bin, axis);
    f3 mn, ext;
    left->prim_start = start;
    left->prim_count = mid - start;
    ext = absolute_extent(tris, NULL, left->prim_start, left->prim_start + left->prim_count, &mn);
    set_bounds(left, mn, ext);
    right->prim_start = mid;
    right->prim_count = end - mid;
    ext = absolute_extent(tris, NULL, right->prim_start, right->prim_start + right->prim_count, &mn);
    set_bounds(right, mn, ext);
}

static void init_node(o_bvh_node *n)                                                        /* BVHNode.cuh:19-25 */
{
    n->is_leaf = 0;
    n->bmin[0] = n->bmin[1] = n->bmin[2] = FLT_MAX;
    n->bmax[0] = n->bmax[1] = n->bmax[2] = -FLT_MAX;
    n->child1 = -1; n->child2 = -1; n->prim_count = 0; n->prim_start = -1;
}

/* BVHBuilder.cu:257-346 */
static void make_partition(o_triangle *tris, int32_t start, int32_t end, int32_t bin_count,
                           o_bvh_node *leftnode, o_bvh_node *rightnode, int32_t *lidx, int32_t *ridx)
{
    float best_bin = 0;
    int best_axis = 0;
    int lowest = INT_MAX;
    f3 mn;
    f3 ext = absolute_extent(tris, NULL, start, end, &mn);
    o_bvh_node parent; init_node(&parent);
    set_bounds(&parent, mn, ext);
    float parent_sa = surface_area(parent.bmin, parent.bmax);
    o_bvh_node left, right;
    init_node(&left); init_node(&right);
    const float mins[3] = { mn.x, mn.y, mn.z }, exts[3] = { ext.x, ext.y, ext.z };
    for (int axis = 0; axis < 3; axis++) {
        float delta = exts[axis] / bin_count;                                               /* :275 */
        for (int i = 1; i < bin_count; i++) {
            float bin = mins[axis] + (i * delta);                                           /* :278 */
            bin_to_shallow_nodes(&left, &right, bin, axis, tris, start, end, lidx, ridx);
            /* int cost = trav_cost + (SA_l/SA_p)*n_l*rayint_cost + (SA_r/SA_p)*n_r*rayint_cost : float sum, truncated */
            float fcost = 1 + ((node_surface_area(&left) / parent_sa) * left.prim_count * 2)
                            + ((node_surface_area(&right) / parent_sa) * right.prim_count * 2);
            int cost = f2i_x86(fcost);
            if (cost < lowest) { lowest = cost; best_axis = axis; best_bin = bin; }
        }
    }
    bin_to_nodes(leftnode, rightnode, best_bin, best_axis, tris, start, end);
}

/* BVHBuilder.cu:11-92 */
int32_t o_bvh_build(o_triangle *tris, int32_t n_tris, int32_t leaf_target, int32_t bin_count,
                    o_bvh_node *nodes, int32_t cap)
{
    if (cap < 1) return -1;
    o_bvh_node root; init_node(&root);
    f3 mn;
    f3 ext = absolute_extent(tris, NULL, 0, n_tris, &mn);
    set_bounds(&root, mn, ext);
    root.prim_start = 0;
    root.prim_count = n_tris;
    if (n_tris <= leaf_target) {                                                            /* :34-43 */
        root.is_leaf = 1;
        nodes[0] = root;
        return 1;
    }
    int32_t *lidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_tris + 1));
    int32_t *ridx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_tris + 1));
    int32_t stack[512];                                 /* MAX_STACK_SIZE, :24; -1 denotes the detached root */
    int sp = 0, count = 0, rc = 0;
    stack[sp++] = -1;
    while (sp > 0) {
        int32_t cur = stack[--sp];
        o_bvh_node *node = cur < 0 ? &root : &nodes[cur];
        if (node->prim_count <= leaf_target) { node->is_leaf = 1; continue; }               /* :54-59 */
        if (count + 3 > cap || sp + 2 > 512) { rc = -1; break; }
        o_bvh_node l, r; init_node(&l); init_node(&r);
        make_partition(tris, node->prim_start, node->prim_start + node->prim_count, bin_count,
                       &l, &r, lidx, ridx);
        if (l.prim_count == 0 || r.prim_count == 0) { rc = -2; break; }   /* reference would loop forever */
        nodes[count] = l; node->child1 = count++;                                           /* :69-70 */
        nodes[count] = r; node->child2 = count++;                                           /* :73-74 */
        stack[sp++] = node->child1;                                                         /* :81-82 */
        stack[sp++] = node->child2;
    }
    free(lidx); free(ridx);
    if (rc < 0) return rc;
    nodes[count++] = root;                                                                  /* :85 */
    return count;
}

/* BVHBuilder.cu:149-173 recursiveBuild: the node's two children are appended after BOTH subtrees (post-order). */
static int recursive_build(o_bvh_node *node, o_triangle *tris, int32_t leaf_target, int32_t bin_count, o_bvh_node *nodes, int32_t cap,
                           int32_t *count, int32_t *lidx, int32_t *ridx, int depth)
{
    if (node->prim_count <= leaf_target) { node->child1 = node->child2 = -1; node->is_leaf = 1; return 0; }     /* :153-158 */
    if (depth > 4096) return -1;
    o_bvh_node l, r; init_node(&l); init_node(&r);
    make_partition(tris, node->prim_start, node->prim_start + node->prim_count, bin_count, &l, &r, lidx, ridx);    /* :164 */
    if (l.prim_count == 0 || r.prim_count == 0) return -2;                 /* the reference would recurse forever */
    int rc = recursive_build(&l, tris, leaf_target, bin_count, nodes, cap, count, lidx, ridx, depth + 1);         /* :166 */
    if (rc < 0) return rc;
    rc = recursive_build(&r, tris, leaf_target, bin_count, nodes, cap, count, lidx, ridx, depth + 1);             /* :167 */
    if (rc < 0) return rc;
    if (*count + 2 > cap) return -1;
    nodes[*count] = l; node->child1 = (*count)++;                                                                  /* :169-170 */
    nodes[*count] = r; node->child2 = (*count)++;                                                                  /* :172-173 */
    return 0;
}

/* BVHBuilder.cu:100-147 BVHBuilder::build: the same partitions as buildIterative, nodes numbered by the recursion. */
int32_t o_bvh_build_recursive(o_triangle *tris, int32_t n_tris, int32_t leaf_target, int32_t bin_count,
                              o_bvh_node *nodes, int32_t cap)
{
    if (cap < 1) return -1;
    o_bvh_node root; init_node(&root);
    f3 mn;
    f3 ext = absolute_extent(tris, NULL, 0, n_tris, &mn);                                   /* :107-110 */
    set_bounds(&root, mn, ext);
    root.prim_count = n_tris;                              /* :112 (primitive_start_idx keeps its default, -1, unless the root is a leaf) */
    if (n_tris <= leaf_target) { root.is_leaf = 1; root.prim_start = 0; nodes[0] = root; return 1; }       /* :115-124 */
    int32_t *lidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_tris + 1));
    int32_t *ridx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_tris + 1));
    int32_t count = 0;
    int rc = 0;
    o_bvh_node l, r; init_node(&l); init_node(&r);
    make_partition(tris, 0, n_tris, bin_count, &l, &r, lidx, ridx);                         /* :129-130 */
    if (l.prim_count == 0 || r.prim_count == 0) rc = -2;
    if (rc == 0) rc = recursive_build(&l, tris, leaf_target, bin_count, nodes, cap - 3, &count, lidx, ridx, 1);   /* :132 */
    if (rc == 0) rc = recursive_build(&r, tris, leaf_target, bin_count, nodes, cap - 3, &count, lidx, ridx, 1);   /* :133 */
    free(lidx); free(ridx);
    if (rc < 0) return rc;
    if (count + 3 > cap) return -1;
    nodes[count] = l; root.child1 = count++;                                                /* :135-136 */
    nodes[count] = r; root.child2 = count++;                                                /* :138-139 */
    nodes[count++] = root;                                                                  /* :142 */
    return count;
}

/* ================= KAT entry points ================= */

void o_kat_random_float(uint32_t seed, int32_t n, float *out, uint32_t *seed_out)
{
    for (int32_t i = 0; i < n; i++) out[i] = random_float(&seed);
    if (seed_out) *seed_out = seed;
}

void o_kat_unit_vec3(const uint32_t *seeds, int32_t n, float *out3, uint32_t *seed_out)
{
    for (int32_t i = 0; i < n; i++) {
        uint32_t s = seeds[i];
        st3(&out3[3 * i], random_unit_vec3(&s));
        seed_out[i] = s;
    }
}

void o_kat_unit_sphere(const uint32_t *seeds, int32_t n, float *out3, uint32_t *seed_out, int32_t *iters)
{
    for (int32_t i = 0; i < n; i++) {
        uint32_t s = seeds[i]; uint64_t it = 0;
        st3(&out3[3 * i], random_unit_sphere_vec3(&s, &it));
        seed_out[i] = s; iters[i] = (int32_t)it;
    }
}

void o_kat_unit_disk(const uint32_t *seeds, int32_t n, float *out2, uint32_t *seed_out)
{
    for (int32_t i = 0; i < n; i++) {
        uint32_t s = seeds[i];
        f2 p = random_in_unit_disk(&s);
        out2[2 * i] = p.x; out2[2 * i + 1] = p.y; seed_out[i] = s;
    }
}

void o_kat_slab(const float *rays6, const float *boxes6, int32_t n, float *out)
{
    for (int32_t i = 0; i < n; i++) {
        ray_t r = make_ray(ld3(&rays6[6 * i]), ld3(&rays6[6 * i + 3]));
        out[i] = slab_intersect(&boxes6[6 * i], &boxes6[6 * i + 3], &r);
    }
}

void o_kat_intersect(const float *rays6, const float *tris9, int32_t n, float *out4, int32_t *hit)
{
    for (int32_t i = 0; i < n; i++) {
        ray_t r = make_ray(ld3(&rays6[6 * i]), ld3(&rays6[6 * i + 3]));
        float p[3][3];
        memcpy(p, &tris9[9 * i], sizeof p);
        short_hit h = tri_intersect(&r, p);
        out4[4 * i] = h.t; out4[4 * i + 1] = h.uvw.x; out4[4 * i + 2] = h.uvw.y; out4[4 * i + 3] = h.uvw.z;
        hit[i] = h.hit;
    }
}

void o_kat_surrounds(const float *min_max_x, int32_t n, int32_t *out)
{
    for (int32_t i = 0; i < n; i++) out[i] = interval_surrounds(min_max_x[3 * i], min_max_x[3 * i + 1], min_max_x[3 * i + 2]);
}

/* Miss (Shaders/Miss.cuh:2-6): a default HitPayload (HitPayload.cuh:8-20) carrying the debug colour; what trace_ray returns
   when the traversal found nothing.  out per case: colour3, hit_distance, has_prim, front_face. */
void o_kat_miss(const float *rays6_color3, int32_t n, float *out4, int32_t *flags2)
{
    for (int32_t i = 0; i < n; i++) {
        hit_payload p;
        memset(&p, 0, sizeof p);
        p.prim = NULL; p.t = -1;                             /* HitPayload.cuh:12,16 defaults */
        p.color = ld3(&rays6_color3[9 * i + 6]);             /* Miss.cuh:4 */
        st3(&out4[4 * i], p.color); out4[4 * i + 3] = p.t;
        flags2[2 * i] = p.prim != NULL; flags2[2 * i + 1] = 1;   /* front_face default: true (HitPayload.cuh:10) */
    }
}

void o_kat_bounds_centroid(const float *boxes6, int32_t n, float *out3)
{
    for (int32_t i = 0; i < n; i++) st3(&out3[3 * i], bounds_centroid(ld3(&boxes6[6 * i]), ld3(&boxes6[6 * i + 3])));
}

void o_kat_surface_area(const float *boxes6, const int32_t *counts, int32_t n, float *out)
{
    for (int32_t i = 0; i < n; i++) {
        o_bvh_node node;
        memset(&node, 0, sizeof node);
        memcpy(node.bmin, &boxes6[6 * i], 12); memcpy(node.bmax, &boxes6[6 * i + 3], 12);
        node.prim_count = counts[i];
        out[i] = node_surface_area(&node);
    }
}

void o_kat_closest_hit(const float *in10, int32_t n, float *out6, int32_t *front)
{
    for (int32_t i = 0; i < n; i++) {
        const float *q = &in10[10 * i];
        ray_t r = make_ray(ld3(&q[0]), ld3(&q[3]));
        f3 pos, nrm;
        front[i] = closest_hit_frame(&r, q[6], ld3(&q[7]), &pos, &nrm);
        st3(&out6[6 * i], pos); st3(&out6[6 * i + 3], nrm);
    }
}

void o_kat_get_ray(const o_camera *cam, const float *uv2, const uint32_t *seeds, int32_t n,
                   float width, float height, float *out6, uint32_t *seed_out)
{
    cam_frame cf = camera_frame(cam, width, height);
    for (int32_t i = 0; i < n; i++) {
        uint32_t s = seeds[i];
        f2 uv = { uv2[2 * i], uv2[2 * i + 1] };
        ray_t r = camera_get_ray(&cf, uv, &s);
        st3(&out6[6 * i], r.orig); st3(&out6[6 * i + 3], r.dir);
        seed_out[i] = s;
    }
}

void o_kat_tex_pixel(const o_texture *tex, const float *uv2, int32_t n, float *out3)
{
    for (int32_t i = 0; i < n; i++) {
        f2 uv = { uv2[2 * i], uv2[2 * i + 1] };
        st3(&out3[3 * i], tex_get_pixel(tex, uv));
    }
}

void o_kat_tex_alpha(const o_texture *tex, const float *uv2, int32_t n, float *out1)
{
    for (int32_t i = 0; i < n; i++) {
        f2 uv = { uv2[2 * i], uv2[2 * i + 1] };
        out1[i] = tex_get_alpha(tex, uv);
    }
}
