/*
 * drt_oracle.h -- CPU restatement of DustRayTracer's megakernel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call it.  The product (dustraytracer_amd/) never does.
 *
 * Parity status: the leaf arithmetic (RNG, camera ray, slab test,
 * Moller-Trumbore, closest-hit frame, texel fetch) is pinned bit-for-bit against a build of the
 * reference's own sources (oracle/_ref/ref_kat, see oracle/Makefile and
 * tests/golden/kat_*.npz).  The control-flow composition (BVH traversal,
 * TraceRay, RayGen, accumulate, BVH builder, glTF flattening) is a restatement
 * with file:line citations and is "parity unpinned": the reference ships no
 * tests or golden images and those translation units need CUDA runtime /
 * thrust / tinygltf, which this image lacks.
 *
 * All citations are relative to /root/reference/DustRayTracer/src/.
 */
#ifndef DRT_ORACLE_H
#define DRT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Core/BVH/BVHNode.cuh:14-44 -- same 44-byte layout as the reference node. */
typedef struct {
    int32_t is_leaf;        /* bool m_IsLeaf + 3 pad bytes */
    float   bmin[3];        /* m_BoundingBox.pMin */
    float   bmax[3];        /* m_BoundingBox.pMax */
    int32_t child1;         /* dev_child1_idx */
    int32_t child2;         /* dev_child2_idx */
    int32_t prim_count;     /* primitives_count */
    int32_t prim_start;     /* primitive_start_idx */
} o_bvh_node;

/* Core/Scene/Triangle.cuh:7-19 + Vertex.cuh:4-12 flattened (no padding games). */
typedef struct {
    float centroid[3];
    float p[3][3];          /* vertex positions */
    float n[3][3];          /* vertex normals (only used while building face_n) */
    float uv[3][2];
    float face_n[3];
    int32_t material;
} o_triangle;

/* Core/Scene/Material.cuh:4-23 -- only the two fields the kernel reads. */
typedef struct {
    float   albedo[3];
    int32_t albedo_tex;     /* AlbedoTextureIndex, -1 = none */
} o_material;

/* Core/Scene/Texture.cuh:4-20 */
typedef struct {
    int32_t width, height, comps;
    int32_t _pad;
    const uint8_t *data;    /* width*height*comps bytes (+ (width+1)*comps zero bytes of padding, see o_tex_get_pixel) */
} o_texture;

/* Core/Scene/RendererSettings.h:4-35 */
typedef struct {
    int32_t gamma_correction;   /* bool */
    int32_t tone_mapping;       /* bool */
    int32_t enable_sunlight;    /* bool */
    int32_t max_samples;
    int32_t ray_bounce_limit;
    int32_t render_mode;        /* 0 NORMALMODE, 1 DEBUGMODE */
    int32_t debug_mode;         /* 0 ALBEDO 1 NORMAL 2 BARYCENTRIC 3 UVS 4 MESHBVH 5 WORLDBVH */
    float   sunlight_dir[2];
    float   sunlight_color[3];
    float   sunlight_intensity;
    float   sky_color[3];
    float   sky_intensity;
} o_settings;

/* Core/Scene/Camera.cuh:30-47 -- the fields GetRay reads. */
typedef struct {
    float exposure;
    float vfov_rad;
    float defocus_angle;
    float focus_dist;
    float position[3];
    float forward[3];
} o_camera;

/* Core/Scene/Material.cuh:9,17,20 -- fields the reference loads (Scene.cu:71-75) and never reads; used only by the opt-in
 * material model below. */
typedef struct {
    float   emissive[3];    /* EmmisiveFactor */
    float   roughness;      /* Roughness */
    int32_t metallic;       /* Metallic (metallicFactor > 0) */
    int32_t transmission;   /* Transmission (Material.cuh:20; the reference's loader never sets it) */
    float   refractive_index; /* refractive_index (Material.cuh:21, default 1.45) */
} o_material_ext;

/* Opt-in extension (NOT reference behaviour; everything 0 / NULL = the reference's image, bit for bit):
 *   emissive  a hit adds  EmmisiveFactor * emissive_scale * throughput  (the throughput before this hit's albedo)
 *   specular  a hit on a Metallic material continues along  reflect(normalize(ray.dir), N) + Roughness * randomUnitSphereVec3
 *             (the same draws as the diffuse bounce; the path ends if that points into the surface) instead of N + ...
 *   transmission  a hit on a Transmission material is a dielectric interface, with the reference's own unused helpers
 *             (Random.cu:26-40): v = normalize(ray.dir), cos = fminf(dot(-v, N), 1), ri = front face ? 1 / refractive_index :
 *             refractive_index; if ri * sqrtf(1 - cos * cos) > 1 (total internal reflection) or reflectance(cos, ri) > randomFloat(seed)
 *             the path continues along reflect(v, N) from P + 0.001 N, else along refract(v, N, ri) from P - 0.001 N (through the
 *             surface).  One randomFloat replaces the bounce's randomUnitSphereVec3; pow(x, 5) is ((x x)(x x)) x.  Takes
 *             precedence over the metallic lobe. */
typedef struct {
    const o_triangle *tris;   int32_t n_tris;
    const o_bvh_node *nodes;  int32_t n_nodes;   /* root = n_nodes-1 (Kernel/TraceRay.cu:20) */
    const o_material *mats;   int32_t n_mats;
    const o_texture  *texs;   int32_t n_texs;
    const o_material_ext *mats_ext;              /* n_mats entries, or NULL */
    int32_t ext_emissive, ext_specular;
    float   ext_emissive_scale;
    int32_t ext_transmission;
} o_scene;

/* Exact work counters; summed over everything rendered. */
typedef struct {
    uint64_t samples;
    uint64_t rays;             /* TraceRay calls */
    uint64_t node_visits;      /* nodes that passed the pop-time culls (heat-map count) */
    uint64_t inner_visits;     /* interior nodes visited (2 child slab tests each) */
    uint64_t tri_tests;        /* Intersection() calls from traverseBVH */
    uint64_t hits_textured;    /* shaded hits whose material has an albedo texture */
    uint64_t hits_flat;
    uint64_t shadow_rays;
    uint64_t inner_visits_shadow;
    uint64_t tri_tests_shadow;
    uint64_t anyhit_alpha;     /* alpha texel fetches in AnyHit */
    uint64_t sphere_iters;     /* randomUnitSphereVec3 loop iterations */
    uint64_t max_stack;        /* max traversal stack height seen */
    uint64_t sphere_iters_traced;   /* of sphere_iters: candidates of directions that a ray is then traced along (the reference also draws one
                                       after the last bounce, RayGen.cuh:88,133-134: its ray is never traced and nothing reads the seed again) */
} o_counters;

void o_default_settings(o_settings *s);
void o_default_camera(o_camera *c);

/* ---- host prep (S1/S2) ---- */

/* Scene.cu:272-302: de-indexed vertex streams (n_tris*3 entries) -> triangles. */
void o_build_triangles(const float *pos, const float *nrm, const float *uv,
                       const int32_t *mat, int32_t n_tris, o_triangle *out);

/* BVHBuilder.cu:11-92 (buildIterative).  tris is reordered in place.
 * nodes_out must hold 2*n_tris+1 nodes.  Returns node count, <0 on error
 * (-2 = builder would not terminate: degenerate partition). */
int32_t o_bvh_build(o_triangle *tris, int32_t n_tris, int32_t leaf_target,
                    int32_t bin_count, o_bvh_node *nodes_out, int32_t nodes_cap);
/* BVHBuilder::build (BVHBuilder.cu:100-173): same tree, nodes in the recursion's order (children after both subtrees, root last) */
int32_t o_bvh_build_recursive(o_triangle *tris, int32_t n_tris, int32_t leaf_target, int32_t bin_count,
                              o_bvh_node *nodes, int32_t cap);

/* ---- render (K1-K14) ---- */

/* Renders frames frame_first .. frame_first+n_frames-1 (frame indices start at
 * 1, RenderKernel.cu:29-34 / Renderer.cu:116) for every row y with
 * (y / stripe_rows) % world == rank.  accum is float3[W*H] (in/out), rgba is
 * float4[W*H] (out, accum/last_frame_index, alpha 1).  n_threads<=0 -> 1. */
void o_render(const o_scene *scene, const o_camera *cam, const o_settings *set,
              int32_t W, int32_t H, uint32_t frame_first, uint32_t n_frames,
              float *accum, float *rgba, int32_t n_threads,
              int32_t stripe_rows, int32_t rank, int32_t world,
              o_counters *counters);

/* Step trace of a pixel rectangle (single-threaded): bytes 'N' pop, 'T' triangle test, 'R' sampler try,
 * 'S' ray shaded, 'P' path done, 'X' pixel done.  For tools/sim_schedule.py. */
size_t o_trace_steps(const o_scene *scene, const o_camera *cam, const o_settings *set, int32_t W, int32_t H,
                     int32_t x0, int32_t y0, int32_t x1, int32_t y1, uint32_t n_frames, unsigned char *buf, size_t cap);

/* ---- known-answer-test entry points for the leaf functions ---- */
uint32_t o_pcg_hash(uint32_t v);                                   /* Random.cu:6-11 */
void o_kat_random_float(uint32_t seed, int32_t n, float *out, uint32_t *seed_out);       /* Random.cu:13-17 */
void o_kat_unit_vec3(const uint32_t *seeds, int32_t n, float *out3, uint32_t *seed_out); /* Random.cu:42-48 */
void o_kat_unit_sphere(const uint32_t *seeds, int32_t n, float *out3, uint32_t *seed_out, int32_t *iters); /* Random.cu:50-58 */
void o_kat_unit_disk(const uint32_t *seeds, int32_t n, float *out2, uint32_t *seed_out); /* Random.cu:60-66 */
/* Bounds.cu:18-41; rays = n x (origin3, dir3) (invDir = 1/dir as Ray.cuh:7), boxes = n x (min3,max3) */
void o_kat_slab(const float *rays6, const float *boxes6, int32_t n, float *out);
/* Intersection.cu:4-36; tris9 = n x (v0,v1,v2); out4 = n x (t,U,V,W), hit[n] */
void o_kat_intersect(const float *rays6, const float *tris9, int32_t n, float *out4, int32_t *hit);
/* BVHNode.cuh:29-35 + Bounds.cu:4-10 (the SAH cost's surface area; 0 for an empty node) */
void o_kat_surface_area(const float *boxes6, const int32_t *counts, int32_t n, float *out);
/* Shaders/ClosestHit.cuh:4-28; in10 = n x (origin3, dir3, t, face_normal3); out6 = n x (position3, normal3), front[n] */
void o_kat_closest_hit(const float *in10, int32_t n, float *out6, int32_t *front);
/* Camera.cu:82-123; uv2 = n x (u,v); out6 = n x (origin3, dir3) */
void o_kat_get_ray(const o_camera *cam, const float *uv2, const uint32_t *seeds, int32_t n,
                   float width, float height, float *out6, uint32_t *seed_out);
/* Texture.cu:33-75 */
void o_kat_tex_pixel(const o_texture *tex, const float *uv2, int32_t n, float *out3);
void o_kat_tex_alpha(const o_texture *tex, const float *uv2, int32_t n, float *out1);

#ifdef __cplusplus
}
#endif
#endif
