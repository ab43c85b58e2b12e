/*
 * drt.h -- C ABI of the MI355X-native DustRayTracer path-tracing core.
 *
 * One shared library (dustraytracer_amd/libdrt_hip.so) exports exactly these
 * symbols.  They are what a binding for the reference's Renderer / Scene /
 * BVHBuilder / Camera classes needs (include/DustRayTracer.hpp is that binding
 * for C++; INTEGRATION.md shows the editor-side change).  Plain pointers and
 * sizes only.  Citations are relative to /root/reference/DustRayTracer/src/.
 *
 * Error model (replaces Editor/Common/CudaCommon.cu:4-13, which prints,
 * cudaDeviceReset()s and exit(99)s): every call returns DRT_OK (0) or a
 * negative drt_status; drt_last_error() returns a thread-local message.
 * Nothing in the library ever calls exit().
 *
 * There is NO CPU fallback: a call that needs the GPU fails with
 * DRT_ERR_DEVICE when no gfx950 device is usable.
 */
#ifndef DRT_H
#define DRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: drt_counters grew by sampler_tries; drt_group_* and drt_material_model entry points (round 2) are part of it */
#define DRT_ABI_VERSION 2

typedef enum {
    DRT_OK = 0,
    DRT_ERR_INVALID = -1,      /* bad argument / bad handle state */
    DRT_ERR_IO = -2,           /* file missing or unreadable */
    DRT_ERR_PARSE = -3,        /* glTF / PNG content not understood */
    DRT_ERR_UNSUPPORTED = -4,  /* valid input outside the reference loader's subset */
    DRT_ERR_DEVICE = -5,       /* HIP error, no device, out of device memory */
    DRT_ERR_BVH = -6           /* builder cannot terminate on this input (the reference hangs) */
} drt_status;

/* ---- PODs mirroring the reference's public data ---- */

/* Core/Scene/RendererSettings.h:4-35 (bools widened to int32 for a stable ABI). */
typedef struct drt_settings {
    int32_t gamma_correction;      /* :22 default 1 */
    int32_t tone_mapping;          /* :23 default 1 */
    int32_t enable_sunlight;       /* :24 default 0 */
    int32_t max_samples;           /* :25 default 500; Render is a no-op once sample_count == max_samples */
    int32_t ray_bounce_limit;      /* :26 default 2; the path loop runs i = 0..limit inclusive */
    int32_t render_mode;           /* :27 0 NORMALMODE, 1 DEBUGMODE */
    int32_t debug_mode;            /* :28 0 ALBEDO 1 NORMAL 2 BARYCENTRIC 3 UVS 4 MESHBVH 5 WORLDBVH */
    float   sunlight_dir[2];       /* :29 */
    float   sunlight_color[3];     /* :30 */
    float   sunlight_intensity;    /* :31 */
    float   sky_color[3];          /* :32 */
    float   sky_intensity;         /* :34 */
} drt_settings;

/* Opt-in material model -- NOT reference behaviour.  The reference loads emissiveFactor, roughnessFactor and
 * metallicFactor (Scene.cu:71-75, Material.cuh:9,17,20) and its kernel reads none of them (RayGen.cuh:111-134).  With everything
 * zero (the default) the image is the reference's, bit for bit.  emissive != 0: a hit adds  EmmisiveFactor * emissive_scale *
 * throughput  (the throughput before that hit's albedo).  specular != 0: a hit on a Metallic material continues along
 * reflect(normalize(ray.dir), N) + Roughness * randomUnitSphereVec3(seed)  -- the same random draws as the diffuse bounce -- and the
 * path ends if that direction points into the surface.  transmission != 0: a hit on a Transmission material (Material.cuh:20-21; the
 * reference's loader never sets it: drt_scene_add_material_ex does) is a dielectric interface, computed with the reference's own
 * unused helpers refract / reflectance (CudaMath/Random.cu:26-40):  v = normalize(ray.dir), cos = fminf(dot(-v, N), 1), ri = front face ?
 * 1 / refractive_index : refractive_index;  total internal reflection (ri * sqrtf(1 - cos^2) > 1) or reflectance(cos, ri) >
 * randomFloat(seed)  ->  continue along reflect(v, N) from P + 0.001 N, else along refract(v, N, ri) from P - 0.001 N.  That one
 * randomFloat replaces the bounce's randomUnitSphereVec3; pow(x, 5) is ((x x)(x x)) x; takes precedence over the metallic lobe.
 * The rule is this library's (the reference has none): oracle/drt_oracle.c states it first, tests/test_material_model.py checks
 * cases that can be derived by hand -- "parity unpinned" by construction.
 * Rendered by path_pool (every lobe) and by the general wave_queue kernel (emissive, specular). */
typedef struct drt_material_model {
    int32_t emissive;
    int32_t specular;
    float   emissive_scale;        /* default 1 */
    int32_t transmission;
} drt_material_model;

/* Core/Scene/Camera.cuh:30-47: the fields the kernel reads (Camera.cu:82-123). */
typedef struct drt_camera {
    float exposure;                /* :32 */
    float vfov_rad;                /* :33 */
    float defocus_angle;           /* :37 */
    float focus_dist;              /* :38 */
    float position[3];             /* :43 m_Position */
    float forward[3];              /* :44 m_Forward_dir */
} drt_camera;

/* Core/Scene/Vertex.cuh:4-12 (32 bytes) */
typedef struct drt_vertex { float position[3]; float normal[3]; float uv[2]; } drt_vertex;

/* Core/Scene/Triangle.cuh:7-19 (128 bytes, same field offsets as the reference struct) */
typedef struct drt_triangle {
    float centroid[3]; float _pad0;
    drt_vertex vertex[3];
    float face_normal[3];
    int32_t material;
} drt_triangle;

/* Core/BVH/BVHNode.cuh:14-44 (44 bytes). The root is the LAST node (BVHBuilder.cu:85). */
typedef struct drt_bvh_node {
    uint8_t is_leaf; uint8_t _pad0[3];
    float bmin[3], bmax[3];
    int32_t child1, child2;
    int32_t prim_count, prim_start;
} drt_bvh_node;

/* Core/Scene/Material.cuh:4-23 (44 bytes); the kernel reads albedo + albedo_tex only (RayGen.cuh:112-117). */
typedef struct drt_material {
    float albedo[3];
    float emissive[3];
    int32_t albedo_tex;
    float roughness;
    uint8_t transmission; uint8_t _pad0[3];
    float refractive_index;
    uint8_t metallic; uint8_t _pad1[3];
} drt_material;

/* Core/Scene/Mesh.cuh:8-18 */
typedef struct drt_mesh { int32_t primitives_offset; int32_t tris_count; } drt_mesh;

typedef struct drt_texture_info { int32_t width, height, components; } drt_texture_info;

/* Exact device-side work counters for one render call (drt_renderer_set_counting). */
typedef struct drt_counters {
    uint64_t samples, rays, node_visits, inner_visits, tri_tests, hits_textured, hits_flat,
             shadow_rays, inner_visits_shadow, tri_tests_shadow;
    /* wave_queue kernel only: executions of the T / N / S / R phase (per wave) and lanes served by them */
    uint64_t phase_execs[4], phase_lanes[4];
    uint64_t phase_ticks[4], wave_ticks;   /* counting build: shader-clock ticks per phase / per wave lifetime, summed over waves */
    uint64_t sampler_tries;                /* path_pool: candidates drawn by randomUnitSphereVec3 (Random.cu:50-58) for bounce directions */
} drt_counters;

typedef struct drt_scene drt_scene;         /* replaces struct Scene, Core/Scene/Scene.cuh:41-57 */
typedef struct drt_renderer drt_renderer;   /* replaces class Renderer, Core/Renderer.hpp:14-47 */

/* ---- library ---- */
int         drt_abi_version(void);
const char *drt_last_error(void);
int         drt_device_count(void);                       /* usable HIP devices, 0 when none */
void        drt_default_settings(drt_settings *out);      /* RendererSettings.h:22-34 */
void        drt_default_camera(drt_camera *out);          /* Camera.cuh:32-46 + EditorLayer.cpp:35-40 */
/* Camera host logic, one implementation for every binding (fp32, the reference's operation order):
 * Camera::Rotate (Camera.cu:61-80; delta = sin_x, cos_x, sin_y, cos_y) updates forward and right in place;
 * Camera::OnUpdate (Camera.cu:44-58) moves position by speed * (right*v.x + up*v.y + forward*v.z) * delta. */
void        drt_camera_rotate(float forward[3], float right[3], const float up[3], const float delta[4]);
void        drt_camera_move(float position[3], const float right[3], const float up[3], const float forward[3],
                            const float velocity[3], float speed, float delta);

/* ---- Scene: host-side load + BVH build (Scene.cu:181-317, BVHBuilder.cu:11-92) ---- */
drt_scene *drt_scene_create(void);
void       drt_scene_destroy(drt_scene *s);                                   /* Scene::~Scene, Scene.cu:319-344 */
int        drt_scene_load_gltf(drt_scene *s, const char *path);               /* Scene::loadGLTFmodel */
/* flags = 0: as above.  DRT_LOAD_STRICT: read the file as the glTF 2.0 specification defines it instead of as the
 * reference's loader does (Scene.cu:120-200 ignores node transforms and the scene graph, accessor byteOffset /
 * componentType / byteStride, reads indices as u16 from byte 0 of the buffer, uses texture indices as image indices, and
 * crashes on nodes without a mesh).  A file that satisfies the reference's assumptions loads identically either way. */
#define DRT_LOAD_STRICT 1u
int        drt_scene_load_gltf_ex(drt_scene *s, const char *path, uint32_t flags);
/* Programmatic alternative to a file: de-indexed streams, 3 vertices per triangle. */
int        drt_scene_set_geometry(drt_scene *s, const float *positions, const float *normals, const float *uvs,
                                  const int32_t *material_ids, int32_t n_tris);
int        drt_scene_add_material(drt_scene *s, const float albedo[3], int32_t albedo_tex);
int        drt_scene_add_material_ex(drt_scene *s, const drt_material *m);   /* every field (emissive, roughness, metallic: drt_material_model) */
/* CudaMath/Random.cu:6-17 on the host: the RNG the kernels use (pcg_hash; randomFloat = hash, then seed / 2^32 in [0, 1]). */
uint32_t   drt_pcg_hash(uint32_t input);
float      drt_random_float(uint32_t *seed);
int        drt_scene_add_texture(drt_scene *s, const uint8_t *texels, int32_t width, int32_t height, int32_t components);
int        drt_scene_build_bvh(drt_scene *s, int32_t target_leaf_prims, int32_t bin_count);  /* BVHBuilder::buildIterative */
/* The same build run on GPU `device`: identical nodes, node order and triangle order -- a bound that
 * is a zero may carry the other sign.  build_ms (may be NULL) receives the device time.  DRT_ERR_DEVICE without a GPU. */
/* BVHBuilder::build (BVH/BVHBuilder.cu:100-173): the same tree and triangle order as drt_scene_build_bvh, with the node array in
 * the order the reference's recursion appends it (a node's two children after both of their subtrees, root last). */
int        drt_scene_build_bvh_recursive(drt_scene *s, int32_t target_leaf_prims, int32_t bin_count);
int        drt_scene_build_bvh_device(drt_scene *s, int32_t target_leaf_prims, int32_t bin_count, int32_t device, float *build_ms);
/* Checks what the kernels index without checks: every triangle's material id, every material's texture index, the BVH's
 * child and triangle ranges.  DRT_ERR_INVALID names the first offender; rendering runs the same check before uploading. */
int        drt_scene_validate(const drt_scene *s);
int32_t    drt_scene_triangle_count(const drt_scene *s);                      /* m_PrimitivesBuffer.size() */
int32_t    drt_scene_node_count(const drt_scene *s);                          /* m_BVHNodes.size() */
int32_t    drt_scene_material_count(const drt_scene *s);
int32_t    drt_scene_texture_count(const drt_scene *s);
int32_t    drt_scene_mesh_count(const drt_scene *s);
int32_t    drt_scene_bvh_depth(const drt_scene *s);                           /* levels, 0 when no BVH */
int        drt_scene_get_triangles(const drt_scene *s, drt_triangle *out, int32_t cap);
int        drt_scene_get_nodes(const drt_scene *s, drt_bvh_node *out, int32_t cap);
int        drt_scene_get_materials(const drt_scene *s, drt_material *out, int32_t cap);
int        drt_scene_get_meshes(const drt_scene *s, drt_mesh *out, int32_t cap);
int        drt_scene_get_texture_info(const drt_scene *s, int32_t index, drt_texture_info *out);
int        drt_scene_get_texture_texels(const drt_scene *s, int32_t index, uint8_t *out, size_t cap);

/* ---- Renderer (Core/Renderer.hpp:14-47, Core/Renderer.cu) ---- */
drt_renderer *drt_renderer_create(int32_t device);                            /* Renderer::Renderer */
void          drt_renderer_destroy(drt_renderer *r);                          /* Renderer::~Renderer */
int           drt_renderer_resize(drt_renderer *r, uint32_t width, uint32_t height);   /* ResizeBuffer: no-op for equal size, else realloc + reset */
int           drt_renderer_set_settings(drt_renderer *r, const drt_settings *s);       /* writes m_RendererSettings (caller resets, EditorLayer.cpp:241-277) */
int           drt_renderer_set_material_model(drt_renderer *r, const drt_material_model *m);   /* see drt_material_model */
int           drt_renderer_get_material_model(const drt_renderer *r, drt_material_model *out);
int           drt_renderer_get_settings(const drt_renderer *r, drt_settings *out);
/* Render: ONE frame index, blocking, returns kernel ms in *delta_ms (Renderer.cu:80-117). No-op when sample_count == max_samples. */
int           drt_renderer_render(drt_renderer *r, const drt_camera *cam, const drt_scene *scene, float *delta_ms);
/* spp batch: frames sample_count .. sample_count+n_frames-1 in one launch; per-pixel sum order ((a+c_f)+c_f+1)+... is kept,
 * so the result is bit-identical to n_frames drt_renderer_render calls.  Clamped so that sample_count never passes max_samples. */
int           drt_renderer_render_batch(drt_renderer *r, const drt_camera *cam, const drt_scene *scene,
                                        uint32_t n_frames, float *delta_ms);
/* Non-blocking form: enqueues the batch on the renderer's stream and returns; drt_renderer_wait blocks until it is done
 * and returns its device time.  Lets a caller keep several frames in flight (one renderer + stream per frame). */
int           drt_renderer_render_batch_async(drt_renderer *r, const drt_camera *cam, const drt_scene *scene, uint32_t n_frames);
int           drt_renderer_wait(drt_renderer *r, float *delta_ms);
int           drt_renderer_reset(drt_renderer *r);                            /* resetAccumulationBuffer: zero + sample_count = 1 */
uint32_t      drt_renderer_width(const drt_renderer *r);                      /* getBufferWidth */
uint32_t      drt_renderer_height(const drt_renderer *r);                     /* getBufferHeight */
uint32_t      drt_renderer_sample_count(const drt_renderer *r);               /* getSampleCount == m_FrameIndex (starts at 1) */
/* Framebuffer out (replaces the GL RGBA32F texture of Renderer.cu:48,70; row 0 = bottom, alpha = 1).
 * Sharded renderers hold local_rows rows; unsharded ones hold all of them. */
uint32_t      drt_renderer_local_rows(const drt_renderer *r);
int           drt_renderer_read_rgba32f(drt_renderer *r, float *dst, size_t dst_floats);   /* width*local_rows*4 */
int           drt_renderer_read_accum(drt_renderer *r, float *dst, size_t dst_floats);     /* width*local_rows*3 */
void         *drt_renderer_device_rgba(drt_renderer *r);                      /* device float4[width*local_rows] */
void         *drt_renderer_device_accum(drt_renderer *r);                     /* device float3[width*local_rows] */

/* ---- multi-GPU sharding (new; the reference is single-device) ---- */
/* This renderer owns the rows y with (y / stripe_rows) % world == rank, stored compactly in stripe order.
 * RNG seeds use the GLOBAL pixel index, so any partition reproduces the single-device image bit for bit. */
int           drt_renderer_set_shard(drt_renderer *r, uint32_t stripe_rows, uint32_t rank, uint32_t world);
/* Use caller-owned device buffers (e.g. torch tensors) instead of internal ones; NULL restores internal. */
int           drt_renderer_bind_buffers(drt_renderer *r, void *device_accum, void *device_rgba);
int           drt_renderer_set_stream(drt_renderer *r, void *hip_stream);     /* hipStream_t, NULL = default stream */
int           drt_renderer_set_counting(drt_renderer *r, int32_t enable);     /* exact work counters (slower kernel) */
int           drt_renderer_get_counters(drt_renderer *r, drt_counters *out);
int           drt_renderer_kernel_info(const drt_renderer *r, char *buf, size_t cap);  /* name/variant of the last kernel */
/* Hint: the caller keeps n launches in flight on this device (several renderers, each on its own stream, driven through
 * drt_renderer_render_batch_async).  Small launches are then given fewer workgroups so that they overlap instead of
 * queueing behind each other.  Default 1 = every launch sized to fill the GPU on its own.  Results do not depend on it. */
int           drt_renderer_set_frames_in_flight(drt_renderer *r, int32_t n);
/* Execution time of the tracing kernel(s) of the last batch that drt_renderer_wait / a blocking render completed, as
 * the kernel itself measured it (first wave in .. last wave out on the device's constant-rate clock).  Unlike the stream
 * events behind *delta_ms it does not include time the launch spent queued behind other streams' work.  0 for the
 * pixel_walk kernel. */
int           drt_renderer_kernel_span(const drt_renderer *r, float *ms);
/* Tracing-kernel launches of the last batch (a batch whose samples exceed the per-launch sample buffer is split). */
int32_t       drt_renderer_launch_count(const drt_renderer *r);
/* rank-0 side of the gather: `gathered` = world shards of padded_rows rows each (as written by the ranks' device_rgba),
 * `image` = full width*height float4.  Runs on `hip_stream`. */
int           drt_assemble_shards(const void *gathered, void *image, uint32_t width, uint32_t height,
                                  uint32_t stripe_rows, uint32_t world, uint32_t padded_rows, void *hip_stream);
/* ---- several GPUs of one node behind one object (one process, N devices) ------------------------------
 * Each device renders its 8-row stripes of the frame (drt_renderer_set_shard; seeds use the global pixel index, so the image
 * is bit-identical to the one-GPU image) on its own stream; the stripes are then gathered into device `devices[0]`'s full
 * RGBA32F image over RCCL -- grouped ncclSend / ncclRecv, every stripe received at its rows of the image (no assemble
 * pass), one xGMI link per peer.  RCCL is loaded (dlopen) only when a group of more than one device is created.
 * Same contract as the renderer otherwise: frame index from 1, no-op at max_samples, blocking render returns wall ms. */
typedef struct drt_group drt_group;
drt_group    *drt_group_create(const int32_t *devices, int32_t n_devices);     /* NULL on failure (drt_last_error) */
void          drt_group_destroy(drt_group *g);
int32_t       drt_group_size(const drt_group *g);
drt_renderer *drt_group_renderer(drt_group *g, int32_t index);                /* the device's renderer (settings, counters, kernel info) */
int           drt_group_resize(drt_group *g, uint32_t width, uint32_t height);
int           drt_group_set_settings(drt_group *g, const drt_settings *s);
int           drt_group_reset(drt_group *g);
int           drt_group_render_batch(drt_group *g, const drt_camera *cam, const drt_scene *scene, uint32_t n_frames, float *delta_ms);
int           drt_group_render_batch_async(drt_group *g, const drt_camera *cam, const drt_scene *scene, uint32_t n_frames);
int           drt_group_wait(drt_group *g, float *delta_ms);
uint32_t      drt_group_sample_count(const drt_group *g);
void         *drt_group_device_rgba(drt_group *g);                             /* float4[width*height] on devices[0] */
int           drt_group_read_rgba32f(drt_group *g, float *dst, size_t dst_floats);
/* Stripe k of `rank` in a `world`-way split: offset in the rank's compact shard, offset in the full image, length -- in floats
 * of an RGBA32F frame.  Returns 0 when the rank has no k-th stripe.  (What the gather's send / receive offsets are made of.) */
int           drt_shard_stripe(uint32_t width, uint32_t height, uint32_t stripe_rows, uint32_t rank, uint32_t world, uint32_t k,
                               uint64_t *src_offset_floats, uint64_t *dst_offset_floats, uint64_t *count_floats);
/* Self-check of the kernels' reciprocal (device_math.hpp exact_rcp) against IEEE 1.0f/x over all 2^32 float bit patterns. */
int           drt_debug_check_rcp(int32_t device, uint64_t *mismatches, uint64_t *fast_path_count);
int           drt_debug_check_sqrt(int32_t device, uint64_t *mismatches, uint64_t *fast_path_count);   /* exact_sqrt vs sqrtf */
/* Host: decode an image file held in memory (PNG, or baseline JPEG) exactly as the loader does for embedded glTF images
 * (Scene.cu:93-114 -> Texture.cu:21-30: native channel count, row 0 first).  `out` may be NULL to query the size only. */
int           drt_debug_decode_image(const uint8_t *file, size_t file_bytes, drt_texture_info *info, uint8_t *out, size_t cap);
/* Device leaf functions on arrays, for known-answer tests against tests/golden/kat_ref.npz.
 * which: 0 unit vec (in u32 seed; out vec3,seed,tries), 1 unit sphere (same), 2 slab (in orig3,dir3,min3,max3; out f32),
 * 3 triangle (in orig3,dir3,v0,v1,v2; out t,U,V,W,hit), 4 camera ray (in u,v,seed; out orig3,dir3,seed; needs cam,width,height),
 * 5 unit disk (in seed; out x,y,seed), 6 closest-hit frame (in orig3,dir3,t,face_normal3; out position3,normal3,front_face). */
int           drt_debug_kat(int32_t device, int32_t which, const void *in, size_t in_bytes, void *out, size_t out_bytes, uint32_t n,
                            const drt_camera *cam, uint32_t width, uint32_t height);
/* Every 32-bit value on a cycle of the RNG hash (Random.cu:6-11) no longer than max_len: (value, length) pairs. */
int           drt_debug_hash_cycles(int32_t device, uint32_t max_len, uint32_t *pairs_out, uint32_t cap_pairs, uint32_t *found);
/* The wave_queue launch packagings (workgroup size, stack entry bytes, triangles per step) this renderer has timed so far, as
 * JSON text: one plan per (kernel, scene shape, view class) with its candidates, their trials and best ns per sample, and
 * the index of the one kept (-1 = still measuring).  All candidates compute the same image. */
int           drt_debug_wave_queue_plans(const drt_renderer *r, char *buf, size_t cap);
/* path_pool kernel statistics of a renderer created with DRT_POOL_STATS=1 in the environment: per queue (N, T0..T3, B, E, R, S)
 * {batches, paths served, shader-clock ticks}, then ticks spent claiming, idle polls, lost claims, wave ticks, claims given up,
 * their ticks, idle ticks. */
int           drt_debug_pool_stats(drt_renderer *r, uint64_t out[40], int32_t reset);
uint32_t      drt_shard_rows(uint32_t height, uint32_t stripe_rows, uint32_t rank, uint32_t world);

#ifdef __cplusplus
}
#endif
#endif
