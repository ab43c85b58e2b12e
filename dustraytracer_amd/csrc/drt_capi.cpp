// drt_capi.cpp -- the C ABI of include/drt.h: scene handles, the renderer (device buffers, per-call
// constants, launches, read-back) and error reporting.
//
// Replaces class Renderer (Core/Renderer.hpp:14-47, Core/Renderer.cu) and InvokeRenderKernel
// (Core/Kernel/RenderKernel.cu:37-58).  No GL interop: the framebuffer is a device float4 array.
#include "../../include/drt.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>

#include "device_scene.hpp"
#include "render_kernels.hpp"
#include "scene_host.hpp"
#include "bvh_build_device.hpp"
#include "png_decode.hpp"

using namespace drt;

namespace {

thread_local std::string g_error;

int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}

int from_exception() {
    try {
        throw;
    } catch (const UnsupportedError &e) { return fail(DRT_ERR_UNSUPPORTED, e.what());
    } catch (const IoError &e) { return fail(DRT_ERR_IO, e.what());
    } catch (const BvhError &e) { return fail(DRT_ERR_BVH, e.what());
    } catch (const DeviceError &e) { return fail(DRT_ERR_DEVICE, e.what());
    } catch (const std::invalid_argument &e) { return fail(DRT_ERR_INVALID, e.what());
    } catch (const std::bad_alloc &) { return fail(DRT_ERR_INVALID, "out of host memory");
    } catch (const std::exception &e) { return fail(DRT_ERR_PARSE, e.what());
    } catch (...) { return fail(DRT_ERR_INVALID, "unknown error"); }
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(DRT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

template <class T>
struct DeviceArray {
    T *ptr = nullptr;
    size_t count = 0;
    hipError_t upload(const std::vector<T> &host) {
        release();
        count = host.size();
        size_t bytes = std::max<size_t>(host.size(), 1) * sizeof(T);
        hipError_t e = hipMalloc((void **)&ptr, bytes);
        if (e != hipSuccess) { ptr = nullptr; return e; }
        if (!host.empty()) e = hipMemcpy(ptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice);
        return e;
    }
    void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; count = 0; }
};

}  // namespace

struct drt_scene {
    HostScene host;
};

struct drt_renderer {
    int device = 0;
    drt_settings settings;
    uint32_t width = 0, height = 0;
    uint32_t frame_index = 1;                 // m_FrameIndex, Renderer.hpp:41
    uint32_t stripe_rows = 1, rank = 0, world = 1, local_rows = 0;
    float *accum = nullptr, *rgba = nullptr;  // internal buffers
    float *ext_accum = nullptr, *ext_rgba = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool counting = false;
    bool pending = false;                     // an asynchronous render was enqueued and not waited for yet
    unsigned long long *counters = nullptr;
    static constexpr int kMaxSpans = 64;
    // One 32-byte record per tracing-kernel launch: {max(~first wave in), max(last wave out), status bits, -}.  The records come
    // zeroed from a ring (one memset per kRecords launches); those of a batch are copied to pinned host memory on the stream, in
    // front of ev_stop, so that drt_renderer_wait reads spans and status without another trip to the device.
    static constexpr int kRecords = 1024;
    unsigned long long *records = nullptr;    // device: kRecords x 4 words
    unsigned long long *records_host = nullptr;   // pinned: the kMaxSpans records of the last batch
    int records_used = 0, batch_first_record = 0;
    int spans_used = 0;                       // launches of the last batch that have a record
    int launches_last = 0;                     // tracing-kernel launches of the last batch (a batch is split by the sample budget)
    float span_ms = 0.f;                       // sum of those spans, filled by drt_renderer_wait
    int wall_clock_khz = 100000;
    static constexpr int kCounters = 256;
    unsigned int *tile_counter = nullptr;     // work queue heads of the tracing kernels: kCounters zeroed blocks (kQueueHeadBlockWords each), one per launch, re-zeroed
    int counters_used = 0;                    // in one memset when all are spent (no memset in front of every launch)
    void *samples = nullptr;                  // wave_queue: one float4 per (pixel, frame) of a launch
    size_t samples_bytes = 0;
    size_t sample_budget = (size_t)1 << 30;   // frames of one batch are split so that a launch needs at most this much
    int num_cus = 256;
    int frames_in_flight = 1;                 // drt_renderer_set_frames_in_flight
    bool use_pixel_walk = false;              // DRT_KERNEL=pixel_walk selects the first (non-persistent) kernel
    int use_path_pool = 1;                    // path_pool where it applies (lean paths, scene in LDS); DRT_KERNEL=wave_queue: never
    bool pool_launched = false;               // the batch in flight launched path_pool at least once
    WaveQueueCache wq_cache;                  // measured choice among wave_queue's launch packagings
    PoolScratch pool_scratch;
    PoolTuning pool_tuning;                   // DRT_POOL_THREADS / _PATHS / _MIN_FILL / _PATIENCE
    uint32_t pool_t_class[3] = { 0, 0, 0 };   // leaf-size classes of the uploaded scene (path_pool's T queues)
    bool scene_has_alpha = false;
    int vote_node = 12, vote_shade = 44, vote_dir = 4, vote_spec = 8;
    int leaf_chain = -1;                              // DRT_LEAF_CHAIN: -1 = by tree depth (<= 4 levels), 0 / 1 = forced
    int vote_tail_node = 4, vote_tail_shade = 36;    // once the queue is empty (DRT_VOTE_TN / DRT_VOTE_TS): pops stop waiting for company   // wave_queue phase-voting thresholds (DRT_VOTE_N/S/R/P override)
    const char *kernel_name = "";
    int launch_shape[5] = { 0, 0, 0, 0, 0 }; // wave_queue: stack slots per lane, workgroups per CU, LDS KiB per workgroup, threads per workgroup; path_pool: + pool paths
    // device copy of the scene last rendered
    const drt_scene *uploaded_scene = nullptr;
    uint64_t uploaded_revision = 0;
    DeviceArray<InnerNode> d_inner;
    DeviceArray<LeafRange> d_leaves;
    DeviceArray<TriHot> d_hot;
    DeviceArray<TriCold> d_cold;
    DeviceArray<MatDev> d_mats;
    DeviceArray<MatExt> d_mats_ext;
    drt_material_model material_model = { 0, 0, 1.0f, 0 };       // emissive, specular, emissive_scale, transmission
    DeviceArray<TexDev> d_texs;
    DeviceArray<uint8_t> d_texels;
    SceneView view;
    int bvh_depth = 0;

    float *cur_accum() const { return ext_accum ? ext_accum : accum; }
    float *cur_rgba() const { return ext_rgba ? ext_rgba : rgba; }
    void free_scene() {
        d_inner.release(); d_leaves.release(); d_hot.release(); d_cold.release();
        d_mats.release(); d_mats_ext.release(); d_texs.release(); d_texels.release();
        uploaded_scene = nullptr;
    }
};

extern "C" {

int drt_abi_version(void) { return DRT_ABI_VERSION; }
int drt_internal_fail(int code, const char *msg) { return fail(code, msg ? msg : ""); }      // for the library's other translation units
const char *drt_last_error(void) { return g_error.c_str(); }

int drt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void drt_default_settings(drt_settings *s) {
    if (!s) return;
    std::memset(s, 0, sizeof *s);
    s->gamma_correction = 1; s->tone_mapping = 1; s->enable_sunlight = 0;
    s->max_samples = 500; s->ray_bounce_limit = 2; s->render_mode = 0; s->debug_mode = 0;
    s->sunlight_dir[0] = -0.803f; s->sunlight_dir[1] = 0.681f;
    s->sunlight_color[0] = 1.000f; s->sunlight_color[1] = 0.944f; s->sunlight_color[2] = 0.917f;
    s->sunlight_intensity = 30;
    s->sky_color[0] = 0.25f; s->sky_color[1] = 0.498f; s->sky_color[2] = 0.80f;
    s->sky_intensity = 20;
}

void drt_default_camera(drt_camera *c) {
    if (!c) return;
    const float PI = 3.14159265359f;             // deg2rad, Camera.cu:125-129
    c->exposure = 1;
    c->vfov_rad = 60 * (PI / 180.f);
    c->defocus_angle = 0;
    c->focus_dist = 10;
    c->position[0] = 0; c->position[1] = 2; c->position[2] = 5;
    c->forward[0] = 0; c->forward[1] = 0; c->forward[2] = -1;
}

// ------------------------------------------------------------------ scene
drt_scene *drt_scene_create(void) {
    try { return new drt_scene(); } catch (...) { from_exception(); return nullptr; }
}
void drt_scene_destroy(drt_scene *s) { delete s; }

int drt_scene_load_gltf(drt_scene *s, const char *path) {
    if (!s || !path) return fail(DRT_ERR_INVALID, "null argument");
    try { s->host.load_gltf(path); return DRT_OK; } catch (...) { return from_exception(); }
}

int drt_scene_load_gltf_ex(drt_scene *s, const char *path, uint32_t flags) {
    if (!s || !path) return fail(DRT_ERR_INVALID, "null argument");
    if (flags & ~DRT_LOAD_STRICT) return fail(DRT_ERR_INVALID, "unknown load flag");
    try { s->host.load_gltf(path, (flags & DRT_LOAD_STRICT) != 0); return DRT_OK; } catch (...) { return from_exception(); }
}

int drt_scene_set_geometry(drt_scene *s, const float *positions, const float *normals, const float *uvs,
                           const int32_t *material_ids, int32_t n_tris) {
    if (!s || n_tris < 0 || (n_tris > 0 && (!positions || !normals || !uvs || !material_ids)))
        return fail(DRT_ERR_INVALID, "null argument");
    try {
        s->host.triangles.clear(); s->host.meshes.clear();
        s->host.set_geometry(positions, normals, uvs, material_ids, n_tris);
        return DRT_OK;
    } catch (...) { return from_exception(); }
}

int drt_scene_add_material(drt_scene *s, const float albedo[3], int32_t albedo_tex) {
    if (!s || !albedo) return fail(DRT_ERR_INVALID, "null argument");
    try {
        drt_material m;
        std::memset(&m, 0, sizeof m);
        std::memcpy(m.albedo, albedo, 12);
        m.albedo_tex = albedo_tex;
        m.refractive_index = 1.45f;
        s->host.materials.push_back(m);
        s->host.revision = HostScene::next_revision();
        return (int)s->host.materials.size() - 1;
    } catch (...) { return from_exception(); }
}

int drt_scene_add_material_ex(drt_scene *s, const drt_material *m) {
    if (!s || !m) return fail(DRT_ERR_INVALID, "null argument");
    try {
        s->host.materials.push_back(*m);
        s->host.revision = HostScene::next_revision();
        return (int)s->host.materials.size() - 1;
    } catch (...) { return from_exception(); }
}

// CudaMath/Random.cu:6-17 on the host (integer hash; the float conversion is exact arithmetic): what a host-side Sampler draws from
uint32_t drt_pcg_hash(uint32_t input) {
    uint32_t state = input * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
float drt_random_float(uint32_t *seed) {
    if (!seed) return 0.f;
    *seed = drt_pcg_hash(*seed);
    return (float)*seed / 4294967296.0f;
}

int drt_scene_add_texture(drt_scene *s, const uint8_t *texels, int32_t width, int32_t height, int32_t components) {
    if (!s || !texels || width <= 0 || height <= 0 || components < 1 || components > 4)
        return fail(DRT_ERR_INVALID, "bad texture");
    try {
        HostTexture t;
        t.width = width; t.height = height; t.components = components;
        t.texels.assign(texels, texels + (size_t)width * height * components);
        s->host.textures.push_back(std::move(t));
        s->host.revision = HostScene::next_revision();
        return (int)s->host.textures.size() - 1;
    } catch (...) { return from_exception(); }
}

int drt_scene_build_bvh(drt_scene *s, int32_t target_leaf_prims, int32_t bin_count) {
    if (!s) return fail(DRT_ERR_INVALID, "null scene");
    try { s->host.build_bvh(target_leaf_prims, bin_count); return DRT_OK; } catch (...) { return from_exception(); }
}

int drt_scene_build_bvh_recursive(drt_scene *s, int32_t target_leaf_prims, int32_t bin_count) {
    if (!s) return fail(DRT_ERR_INVALID, "null scene");
    try { s->host.build_bvh(target_leaf_prims, bin_count); s->host.renumber_as_recursive_build(); return DRT_OK; } catch (...) { return from_exception(); }
}

int drt_scene_build_bvh_device(drt_scene *s, int32_t target_leaf_prims, int32_t bin_count, int32_t device, float *build_ms) {
    if (!s) return fail(DRT_ERR_INVALID, "null scene");
    if (build_ms) *build_ms = 0.f;
    try {
        const float ms = s->host.build_bvh_on_device(target_leaf_prims, bin_count, device);
        if (build_ms) *build_ms = ms;
        return DRT_OK;
    } catch (...) { return from_exception(); }
}

int drt_scene_validate(const drt_scene *s) {
    if (!s) return fail(DRT_ERR_INVALID, "null scene");
    try { (void)s->host.pack(); return DRT_OK; } catch (...) { return from_exception(); }
}

int32_t drt_scene_triangle_count(const drt_scene *s) { return s ? (int32_t)s->host.triangles.size() : 0; }
int32_t drt_scene_node_count(const drt_scene *s) { return s ? (int32_t)s->host.nodes.size() : 0; }
int32_t drt_scene_material_count(const drt_scene *s) { return s ? (int32_t)s->host.materials.size() : 0; }
int32_t drt_scene_texture_count(const drt_scene *s) { return s ? (int32_t)s->host.textures.size() : 0; }
int32_t drt_scene_mesh_count(const drt_scene *s) { return s ? (int32_t)s->host.meshes.size() : 0; }
int32_t drt_scene_bvh_depth(const drt_scene *s) { return s ? s->host.bvh_depth() : 0; }

#define DRT_COPY_OUT(vec)                                                                          \
    if (!s || (!out && cap > 0) || cap < 0) return fail(DRT_ERR_INVALID, "bad argument");          \
    {                                                                                              \
        size_t n = std::min<size_t>((size_t)cap, s->host.vec.size());                              \
        if (n) std::memcpy(out, s->host.vec.data(), n * sizeof(s->host.vec[0]));                   \
        return (int)n;                                                                             \
    }

int drt_scene_get_triangles(const drt_scene *s, drt_triangle *out, int32_t cap) { DRT_COPY_OUT(triangles) }
int drt_scene_get_nodes(const drt_scene *s, drt_bvh_node *out, int32_t cap) { DRT_COPY_OUT(nodes) }
int drt_scene_get_materials(const drt_scene *s, drt_material *out, int32_t cap) { DRT_COPY_OUT(materials) }
int drt_scene_get_meshes(const drt_scene *s, drt_mesh *out, int32_t cap) { DRT_COPY_OUT(meshes) }

int drt_scene_get_texture_info(const drt_scene *s, int32_t index, drt_texture_info *out) {
    if (!s || !out || index < 0 || (size_t)index >= s->host.textures.size()) return fail(DRT_ERR_INVALID, "bad texture index");
    const HostTexture &t = s->host.textures[(size_t)index];
    out->width = t.width; out->height = t.height; out->components = t.components;
    return DRT_OK;
}

int drt_scene_get_texture_texels(const drt_scene *s, int32_t index, uint8_t *out, size_t cap) {
    if (!s || !out || index < 0 || (size_t)index >= s->host.textures.size()) return fail(DRT_ERR_INVALID, "bad texture index");
    const HostTexture &t = s->host.textures[(size_t)index];
    if (cap < t.texels.size()) return fail(DRT_ERR_INVALID, "destination too small");
    std::memcpy(out, t.texels.data(), t.texels.size());
    return DRT_OK;
}

// ------------------------------------------------------------------ renderer
uint32_t drt_shard_rows(uint32_t height, uint32_t stripe_rows, uint32_t rank, uint32_t world) {
    if (stripe_rows == 0 || world == 0 || rank >= world) return 0;
    uint32_t stripes = (height + stripe_rows - 1) / stripe_rows, rows = 0;
    for (uint32_t s = rank; s < stripes; s += world)
        rows += std::min(stripe_rows, height - s * stripe_rows);
    return rows;
}

static int realloc_buffers(drt_renderer *r) {
    HIP_TRY(hipSetDevice(r->device));
    if (r->accum) { (void)hipFree(r->accum); r->accum = nullptr; }
    if (r->rgba) { (void)hipFree(r->rgba); r->rgba = nullptr; }
    r->local_rows = drt_shard_rows(r->height, r->stripe_rows, r->rank, r->world);
    size_t px = std::max<size_t>((size_t)r->width * r->local_rows, 1);
    HIP_TRY(hipMalloc((void **)&r->accum, px * 3 * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&r->rgba, px * 4 * sizeof(float)));
    HIP_TRY(hipMemset(r->rgba, 0, px * 4 * sizeof(float)));
    return DRT_OK;
}

drt_renderer *drt_renderer_create(int32_t device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        fail(DRT_ERR_DEVICE, "no usable HIP device: this library has no CPU fallback");
        return nullptr;
    }
    if (device < 0 || device >= n) { fail(DRT_ERR_INVALID, "device index out of range"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { fail(DRT_ERR_DEVICE, "hipSetDevice failed"); return nullptr; }
    drt_renderer *r = nullptr;
    try { r = new drt_renderer(); } catch (...) { from_exception(); return nullptr; }
    r->device = device;
    drt_default_settings(&r->settings);
    const char *which = std::getenv("DRT_KERNEL");
    r->use_pixel_walk = which && std::strcmp(which, "pixel_walk") == 0;
    if (r->use_pixel_walk && !pixel_walk_built_in()) {
        fail(DRT_ERR_UNSUPPORTED, "DRT_KERNEL=pixel_walk: that kernel is not part of this library (the tests build libdrt_hip_pixel_walk.so: make pixel-walk)");
        delete r;
        return nullptr;
    }
    r->use_path_pool = !(which && (std::strcmp(which, "wave_queue") == 0 || std::strcmp(which, "pixel_walk") == 0));
    auto env_int = [](const char *name, int dflt) { const char *v = std::getenv(name); return (v && *v) ? std::atoi(v) : dflt; };
    r->vote_node = std::max(1, env_int("DRT_VOTE_N", r->vote_node));
    r->vote_shade = std::max(1, env_int("DRT_VOTE_S", r->vote_shade));
    r->vote_dir = std::max(1, env_int("DRT_VOTE_R", r->vote_dir));
    r->vote_spec = std::max(1, env_int("DRT_VOTE_P", r->vote_spec));
    r->vote_tail_node = std::max(1, env_int("DRT_VOTE_TN", r->vote_tail_node));
    r->vote_tail_shade = std::max(1, env_int("DRT_VOTE_TS", r->vote_tail_shade));
    r->leaf_chain = env_int("DRT_LEAF_CHAIN", r->leaf_chain);
    r->sample_budget = (size_t)std::max(1, env_int("DRT_SAMPLE_MB", 1024)) << 20;
    r->pool_tuning.threads = env_int("DRT_POOL_THREADS", 0); r->pool_tuning.paths = env_int("DRT_POOL_PATHS", 0);
    r->pool_tuning.stack_lds = env_int("DRT_POOL_STACK_LDS", r->pool_tuning.stack_lds);
    r->pool_tuning.min_fill = env_int("DRT_POOL_MIN_FILL", r->pool_tuning.min_fill);
    r->pool_tuning.patience = env_int("DRT_POOL_PATIENCE", r->pool_tuning.patience);
    r->pool_tuning.n_loop = env_int("DRT_POOL_N_LOOP", r->pool_tuning.n_loop);
    r->pool_tuning.n_min_lanes = env_int("DRT_POOL_N_MIN", r->pool_tuning.n_min_lanes);
    r->pool_tuning.n_fuse_loop = env_int("DRT_POOL_N_FUSE_LOOP", r->pool_tuning.n_fuse_loop);
    r->pool_tuning.n_fuse_min = env_int("DRT_POOL_N_FUSE_MIN", r->pool_tuning.n_fuse_min);
    r->pool_tuning.cold_lds_kb = env_int("DRT_POOL_COLD_KB", r->pool_tuning.cold_lds_kb);
    r->pool_tuning.share_grid = env_int("DRT_POOL_SHARE_GRID", r->pool_tuning.share_grid);
    r->pool_tuning.dir_tries = env_int("DRT_POOL_DIR_TRIES", r->pool_tuning.dir_tries);
    if (env_int("DRT_POOL_STATS", 0) != 0 && hipMalloc((void **)&r->pool_tuning.stats, 40 * sizeof(unsigned long long)) == hipSuccess)
        (void)hipMemset(r->pool_tuning.stats, 0, 40 * sizeof(unsigned long long));
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) r->num_cus = cus;
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) r->wall_clock_khz = khz;
    if (hipEventCreate(&r->ev_start) != hipSuccess || hipEventCreate(&r->ev_stop) != hipSuccess ||
        hipMalloc((void **)&r->counters, sizeof(drt_counters)) != hipSuccess ||
        hipMalloc((void **)&r->records, sizeof(unsigned long long) * 4 * drt_renderer::kRecords) != hipSuccess ||
        hipMemset(r->records, 0, sizeof(unsigned long long) * 4 * drt_renderer::kRecords) != hipSuccess ||
        hipHostMalloc((void **)&r->records_host, sizeof(unsigned long long) * 4 * drt_renderer::kMaxSpans, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void **)&r->tile_counter, sizeof(unsigned int) * drt_renderer::kCounters * kQueueHeadBlockWords) != hipSuccess ||
        hipMemset(r->tile_counter, 0, sizeof(unsigned int) * drt_renderer::kCounters * kQueueHeadBlockWords) != hipSuccess) {
        fail(DRT_ERR_DEVICE, "cannot create HIP events / counter buffer");
        drt_renderer_destroy(r);
        return nullptr;
    }
    return r;
}

void drt_renderer_destroy(drt_renderer *r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    r->free_scene();
    if (r->accum) (void)hipFree(r->accum);
    if (r->rgba) (void)hipFree(r->rgba);
    if (r->counters) (void)hipFree(r->counters);
    if (r->records) (void)hipFree(r->records);
    if (r->records_host) (void)hipHostFree(r->records_host);
    if (r->tile_counter) (void)hipFree(r->tile_counter);
    if (r->pool_tuning.stats) (void)hipFree(r->pool_tuning.stats);
    if (r->pool_scratch.aux) (void)hipFree(r->pool_scratch.aux);
    if (r->pool_scratch.aux_slot) (void)hipFree(r->pool_scratch.aux_slot);
    if (r->pool_scratch.aux_light) (void)hipFree(r->pool_scratch.aux_light);
    if (r->pool_scratch.aux_next) (void)hipFree(r->pool_scratch.aux_next);
    if (r->pool_scratch.aux_stack) (void)hipFree(r->pool_scratch.aux_stack);
    if (r->samples) (void)hipFree(r->samples);
    if (r->ev_start) (void)hipEventDestroy(r->ev_start);
    if (r->ev_stop) (void)hipEventDestroy(r->ev_stop);
    delete r;
}

int drt_renderer_reset(drt_renderer *r) {                      // Renderer.cu:132-136
    if (!r) return fail(DRT_ERR_INVALID, "null renderer");
    HIP_TRY(hipSetDevice(r->device));
    if (r->cur_accum() && r->width && r->local_rows)
        HIP_TRY(hipMemsetAsync(r->cur_accum(), 0, (size_t)r->width * r->local_rows * 3 * sizeof(float), r->stream));
    r->frame_index = 1;
    return DRT_OK;
}

int drt_renderer_resize(drt_renderer *r, uint32_t width, uint32_t height) {   // Renderer.cu:29-78
    if (!r) return fail(DRT_ERR_INVALID, "null renderer");
    if (width == r->width && height == r->height) return DRT_OK;
    if ((uint64_t)width * height > (1ull << 31)) return fail(DRT_ERR_INVALID, "framebuffer too large (pixel index is 32-bit, RayGen.cuh:74)");
    if (r->ext_accum || r->ext_rgba) return fail(DRT_ERR_INVALID, "unbind external buffers before resizing");
    r->width = width; r->height = height;
    int rc = realloc_buffers(r);
    if (rc != DRT_OK) return rc;
    return drt_renderer_reset(r);
}

int drt_renderer_set_shard(drt_renderer *r, uint32_t stripe_rows, uint32_t rank, uint32_t world) {
    if (!r || stripe_rows == 0 || world == 0 || rank >= world) return fail(DRT_ERR_INVALID, "bad shard description");
    if (r->ext_accum || r->ext_rgba) return fail(DRT_ERR_INVALID, "unbind external buffers before re-sharding");
    r->stripe_rows = stripe_rows; r->rank = rank; r->world = world;
    if (r->width && r->height) {
        int rc = realloc_buffers(r);
        if (rc != DRT_OK) return rc;
        return drt_renderer_reset(r);
    }
    return DRT_OK;
}

int drt_renderer_bind_buffers(drt_renderer *r, void *device_accum, void *device_rgba) {
    if (!r) return fail(DRT_ERR_INVALID, "null renderer");
    if ((device_accum == nullptr) != (device_rgba == nullptr)) return fail(DRT_ERR_INVALID, "bind both buffers or neither");
    r->ext_accum = (float *)device_accum;
    r->ext_rgba = (float *)device_rgba;
    return DRT_OK;
}

int drt_renderer_set_stream(drt_renderer *r, void *hip_stream) {
    if (!r) return fail(DRT_ERR_INVALID, "null renderer");
    if (r->stream != (hipStream_t)hip_stream) {
        // launches still running on the old stream own queue-head counters that the new stream's bulk re-zero must not touch
        HIP_TRY(hipSetDevice(r->device));
        HIP_TRY(hipStreamSynchronize(r->stream));
    }
    r->stream = (hipStream_t)hip_stream;
    return DRT_OK;
}

int drt_renderer_set_settings(drt_renderer *r, const drt_settings *s) {
    if (!r || !s) return fail(DRT_ERR_INVALID, "null argument");
    r->settings = *s;
    return DRT_OK;
}
int drt_renderer_set_material_model(drt_renderer *r, const drt_material_model *m) {
    if (!r || !m) return fail(DRT_ERR_INVALID, "null argument");
    r->material_model = *m;
    return DRT_OK;
}
int drt_renderer_get_material_model(const drt_renderer *r, drt_material_model *out) {
    if (!r || !out) return fail(DRT_ERR_INVALID, "null argument");
    *out = r->material_model;
    return DRT_OK;
}
int drt_renderer_get_settings(const drt_renderer *r, drt_settings *out) {
    if (!r || !out) return fail(DRT_ERR_INVALID, "null argument");
    *out = r->settings;
    return DRT_OK;
}

uint32_t drt_renderer_width(const drt_renderer *r) { return r ? r->width : 0; }
uint32_t drt_renderer_height(const drt_renderer *r) { return r ? r->height : 0; }
uint32_t drt_renderer_sample_count(const drt_renderer *r) { return r ? r->frame_index : 0; }
uint32_t drt_renderer_local_rows(const drt_renderer *r) { return r ? r->local_rows : 0; }
void *drt_renderer_device_rgba(drt_renderer *r) { return r ? r->cur_rgba() : nullptr; }
void *drt_renderer_device_accum(drt_renderer *r) { return r ? r->cur_accum() : nullptr; }

int drt_renderer_set_counting(drt_renderer *r, int32_t enable) {
    if (!r) return fail(DRT_ERR_INVALID, "null renderer");
    r->counting = enable != 0;
    return DRT_OK;
}

int drt_renderer_get_counters(drt_renderer *r, drt_counters *out) {
    if (!r || !out) return fail(DRT_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipStreamSynchronize(r->stream));
    HIP_TRY(hipMemcpy(out, r->counters, sizeof *out, hipMemcpyDeviceToHost));
    return DRT_OK;
}

// Camera.cu:61-80.  helper_math.cuh: cross(a,b) = (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x),
// dot = a.x*b.x + a.y*b.y + a.z*b.z; every product and sum rounds on its own (-ffp-contract=off).
static void rotate_about(float v[3], const float k[3], float s, float c) {
    const float kxv[3] = { k[1] * v[2] - k[2] * v[1], k[2] * v[0] - k[0] * v[2], k[0] * v[1] - k[1] * v[0] };
    const float d = k[0] * v[0] + k[1] * v[1] + k[2] * v[2];
    const float one_minus_c = 1 - c;
    float r[3];
    for (int i = 0; i < 3; i++) r[i] = ((v[i] * c) + (kxv[i] * s)) + ((k[i] * d) * one_minus_c);
    for (int i = 0; i < 3; i++) v[i] = r[i];
}

void drt_camera_rotate(float forward[3], float right[3], const float up[3], const float delta[4]) {
    if (!forward || !right || !up || !delta) return;
    rotate_about(forward, up, delta[0], delta[1]);
    rotate_about(forward, right, delta[2], delta[3]);
    const float r[3] = { forward[1] * up[2] - forward[2] * up[1], forward[2] * up[0] - forward[0] * up[2],
                         forward[0] * up[1] - forward[1] * up[0] };
    for (int i = 0; i < 3; i++) right[i] = r[i];
}

// Camera.cu:44-58 (glm::mat3 * vec3 = col0*v.x + col1*v.y + col2*v.z, then m_Position += speed * move * delta)
void drt_camera_move(float position[3], const float right[3], const float up[3], const float forward[3],
                     const float velocity[3], float speed, float delta) {
    if (!position || !right || !up || !forward || !velocity) return;
    for (int i = 0; i < 3; i++) {
        const float move = (right[i] * velocity[0] + up[i] * velocity[1]) + forward[i] * velocity[2];
        position[i] = position[i] + (speed * move) * delta;
    }
}

int drt_renderer_set_frames_in_flight(drt_renderer *r, int32_t n) {
    if (!r || n < 1) return fail(DRT_ERR_INVALID, "bad argument");
    r->frames_in_flight = n;
    return DRT_OK;
}

int drt_renderer_kernel_span(const drt_renderer *r, float *ms) {
    if (!r || !ms) return fail(DRT_ERR_INVALID, "bad argument");
    *ms = r->span_ms;
    return DRT_OK;
}

int32_t drt_renderer_launch_count(const drt_renderer *r) { return r ? r->launches_last : 0; }

int drt_renderer_kernel_info(const drt_renderer *r, char *buf, size_t cap) {
    if (!r || !buf || cap == 0) return fail(DRT_ERR_INVALID, "bad argument");
    if (r->launch_shape[1] > 0 && std::strncmp(r->kernel_name, "path_pool", 9) == 0)
    {
        // (launch_shape[0]: stack levels | levels kept in LDS << 8)
        const int levels = r->launch_shape[0] & 255, in_lds = r->launch_shape[0] >> 8;
        char stack[48];
        if (in_lds < levels) std::snprintf(stack, sizeof stack, "%d(%d in LDS)", levels, in_lds);
        else std::snprintf(stack, sizeof stack, "%d", levels);
        std::snprintf(buf, cap, "%s stack=%s wg/CU=%d threads=%d paths=%d lds=%dKiB", r->kernel_name, stack, r->launch_shape[1],
                      r->launch_shape[3], r->launch_shape[4], r->launch_shape[2]);
    }
    else if (r->launch_shape[1] > 0 && std::strncmp(r->kernel_name, "wave_queue", 10) == 0)
        std::snprintf(buf, cap, "%s stack=%d%s wg/CU=%d%s lds=%dKiB", r->kernel_name, r->launch_shape[0], (r->launch_shape[3] & 1) ? "x6B" : ((r->launch_shape[3] & 2) ? " tris=3" : ""),
                      r->launch_shape[1], r->launch_shape[3] >= 512 ? "x512" : "", r->launch_shape[2]);
    else
        std::snprintf(buf, cap, "%s", r->kernel_name);
    return DRT_OK;
}

static int upload_scene(drt_renderer *r, const drt_scene *scene) {
    if (r->uploaded_scene == scene && r->uploaded_revision == scene->host.revision) return DRT_OK;
    PackedScene ps;
    try { ps = scene->host.pack(); } catch (...) { return from_exception(); }
    r->free_scene();
    HIP_TRY(r->d_inner.upload(ps.inner));
    HIP_TRY(r->d_leaves.upload(ps.leaves));
    HIP_TRY(r->d_hot.upload(ps.tri_hot));
    HIP_TRY(r->d_cold.upload(ps.tri_cold));
    HIP_TRY(r->d_mats.upload(ps.mats));
    HIP_TRY(r->d_mats_ext.upload(ps.mats_ext));
    HIP_TRY(r->d_texs.upload(ps.texs));
    HIP_TRY(r->d_texels.upload(ps.texels));
    SceneView &v = r->view;
    v.inner = r->d_inner.ptr; v.leaves = r->d_leaves.ptr; v.tri_hot = r->d_hot.ptr; v.tri_cold = r->d_cold.ptr;
    v.mats = r->d_mats.ptr; v.mats_ext = r->d_mats_ext.ptr; v.texs = r->d_texs.ptr; v.texels = r->d_texels.ptr;
    v.n_inner = (uint32_t)ps.inner.size(); v.n_leaves = (uint32_t)ps.leaves.size();
    v.n_tris = (uint32_t)ps.tri_hot.size(); v.n_mats = (uint32_t)ps.mats.size(); v.n_texs = (uint32_t)ps.texs.size();
    v.root_ref = ps.root_ref;
    std::memcpy(v.root_min, ps.root_min, 12);
    std::memcpy(v.root_max, ps.root_max, 12);
    r->bvh_depth = ps.depth;
    r->scene_has_alpha = ps.any_alpha_texture;
    path_pool_leaf_classes(ps.leaves, r->pool_t_class);
    if (const char *e = std::getenv("DRT_POOL_T_CLASSES")) {              // experiments: "a,b,c" = upper step counts of T0, T1, T2
        unsigned a0 = 0, a1 = 0, a2 = 0;
        if (std::sscanf(e, "%u,%u,%u", &a0, &a1, &a2) == 3) { r->pool_t_class[0] = a0; r->pool_t_class[1] = a1; r->pool_t_class[2] = a2; }
    }
    if (std::getenv("DRT_POOL_VERBOSE")) std::fprintf(stderr, "path_pool leaf classes: %u %u %u\n", r->pool_t_class[0], r->pool_t_class[1], r->pool_t_class[2]);
    r->uploaded_scene = scene;
    r->uploaded_revision = scene->host.revision;
    return DRT_OK;
}

// Per-frame constants of Camera::GetRay (Camera.cu:84-103) and RayGen (RayGen.cuh:68-72), computed on the
// host with the same fp32 operations in the same order (host libm for tan/sin/cos).
static void fill_frame_params(const drt_renderer *r, const drt_camera *cam, FrameParams &fp) {
    const drt_settings &s = r->settings;
    const float width = (float)r->width, height = (float)r->height;       // Camera.cu:82 takes floats
    float theta = cam->vfov_rad / 2;
    float fov_factor = tanf(theta / 2.0f);
    float aspect_ratio = width / height;
    float plane_h = 2.0f * fov_factor * cam->focus_dist;
    float plane_w = plane_h * aspect_ratio;
    V3 forward_dir = normalize(V3{ cam->forward[0], cam->forward[1], cam->forward[2] });
    V3 right_dir = normalize(cross(forward_dir, V3{ 0, 1, 0 }));
    V3 up_dir = cross(right_dir, forward_dir);
    V3 horizontal = plane_w * right_dir, vertical = plane_h * up_dir;
    const float PI = 3.14159265359f;
    float defocus_radius = cam->focus_dist * tanf((cam->defocus_angle * (PI / 180.f)) / 2.0f);
    V3 disk_u = defocus_radius * right_dir, disk_v = defocus_radius * up_dir;
    V3 fwd_focus = forward_dir * cam->focus_dist;
    auto put = [](float *dst, V3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; };
    std::memcpy(fp.cam_pos, cam->position, 12);
    put(fp.fwd_focus, fwd_focus); put(fp.horizontal, horizontal); put(fp.vertical, vertical);
    put(fp.disk_u, disk_u); put(fp.disk_v, disk_v);
    fp.defocus = !(cam->defocus_angle <= 0);
    fp.exposure = cam->exposure;

    float sx = sinf(s.sunlight_dir[0]), sy = sinf(s.sunlight_dir[1]), cx = cosf(s.sunlight_dir[0]);
    put(fp.sunpos, V3{ sx * (1 - sy), sy, cx * (1 - sy) } * 100.0f);
    put(fp.suncol, V3{ s.sunlight_color[0], s.sunlight_color[1], s.sunlight_color[2] } * s.sunlight_intensity);
    std::memcpy(fp.sky_color, s.sky_color, 12);
    fp.sky_intensity = s.sky_intensity;
    fp.gamma_correction = s.gamma_correction != 0; fp.tone_mapping = s.tone_mapping != 0;
    fp.enable_sunlight = s.enable_sunlight != 0;
    fp.bounce_limit = s.ray_bounce_limit;
    fp.render_mode = s.render_mode; fp.debug_mode = s.debug_mode;
    fp.ext_emissive = r->material_model.emissive != 0; fp.ext_specular = r->material_model.specular != 0;
    fp.ext_emissive_scale = r->material_model.emissive_scale;
    fp.ext_transmission = r->material_model.transmission != 0;
    fp.width = r->width; fp.height = r->height;
    fp.stripe_rows = r->stripe_rows; fp.rank = r->rank; fp.world = r->world; fp.local_rows = r->local_rows;
    fp.accum = r->cur_accum(); fp.rgba = r->cur_rgba();
    fp.counters = r->counting ? r->counters : nullptr;
    fp.vote_node = r->vote_node; fp.vote_shade = r->vote_shade; fp.vote_dir = r->vote_dir; fp.vote_spec = r->vote_spec; fp.frames_in_flight = r->frames_in_flight; fp.vote_tail_node = r->vote_tail_node; fp.vote_tail_shade = r->vote_tail_shade;
    {   // tile rows are visited with a golden-ratio stride (kernel_wave_queue.hip, DRT_CHUNK_ORDER)
        const uint32_t tiles_y = (r->local_rows + 7) / 8;
        uint32_t step = 1;
        if (tiles_y > 2 && tiles_y < 65536) {
            step = std::max<uint32_t>(1, (uint32_t)(0.6180339887 * tiles_y + 0.5));
            auto gcd = [](uint32_t a, uint32_t b) { while (b) { uint32_t t = a % b; a = b; b = t; } return a; };
            while (gcd(step, tiles_y) != 1) step++;
        }
        fp.row_step = step;
    }
    fp.leaf_chain = r->leaf_chain < 0 ? (r->bvh_depth <= 4 ? 1 : 0) : (r->leaf_chain != 0);
}

static int render_batch_impl(drt_renderer *r, const drt_camera *cam, const drt_scene *scene, uint32_t n_frames,
                             float *delta_ms, bool blocking) {
    if (!r || !cam || !scene) return fail(DRT_ERR_INVALID, "null argument");
    if (delta_ms) *delta_ms = 0.f;
    r->pending = false;
    r->pool_launched = false;
    if (r->width == 0 || r->height == 0) return fail(DRT_ERR_INVALID, "ResizeBuffer has not been called");
    // Renderer.cu:82: nothing happens once m_FrameIndex == max_samples, so at most max_samples-1 frames accumulate.
    if ((int64_t)r->frame_index == (int64_t)r->settings.max_samples) return DRT_OK;
    if (r->settings.max_samples > 0 && r->frame_index < (uint32_t)r->settings.max_samples)
        n_frames = std::min<uint32_t>(n_frames, (uint32_t)r->settings.max_samples - r->frame_index);
    if (n_frames == 0) return DRT_OK;
    HIP_TRY(hipSetDevice(r->device));
    // hipGetLastError() after a launch reports the last error of ANY earlier runtime call of this thread -- also one that another
    // library made and handled (RCCL probing peers answers "invalid device ordinal" on a one-GPU box): start from a clean slate
    (void)hipGetLastError();
    int rc = upload_scene(r, scene);
    if (rc != DRT_OK) return rc;
    if (r->bvh_depth > 64) return fail(DRT_ERR_UNSUPPORTED, "BVH deeper than 64 levels (the reference's traversal stack, BVHTraversal.cuh:17)");

    FrameParams fp;
    std::memset(&fp, 0, sizeof fp);
    fill_frame_params(r, cam, fp);
    fp.frame_first = r->frame_index;
    fp.n_frames = n_frames;
    if (r->counting) HIP_TRY(hipMemsetAsync(r->counters, 0, sizeof(drt_counters), r->stream));

    r->spans_used = 0;
    HIP_TRY(hipEventRecord(r->ev_start, r->stream));                   // Renderer.cu:97
    const bool material_ext = fp.ext_emissive || fp.ext_specular;          // only the general wave_queue kernel implements it
    if (r->use_pixel_walk && !material_ext) {
        HIP_TRY(launch_render(r->view, fp, r->bvh_depth, r->counting, r->stream, &r->kernel_name));
        r->launches_last = 1;
    } else {
        // split the batch so that the per-sample colour buffer of one launch stays within the budget
        const size_t per_frame = (size_t)r->width * r->local_rows * 4 * sizeof(float);
        uint32_t frames_per_launch = (uint32_t)std::max<size_t>(1, std::min<size_t>(n_frames, r->sample_budget / std::max<size_t>(per_frame, 1)));
        const size_t need = per_frame * frames_per_launch;
        if (need > r->samples_bytes) {
            if (r->samples) { (void)hipFree(r->samples); r->samples = nullptr; r->samples_bytes = 0; }
            HIP_TRY(hipMalloc(&r->samples, std::max<size_t>(need, 16)));
            r->samples_bytes = need;
        }
        r->spans_used = 0;
        r->launches_last = 0;
        if (r->records_used + drt_renderer::kMaxSpans > drt_renderer::kRecords) {      // stream order: every launch that used them is over by then
            HIP_TRY(hipMemsetAsync(r->records, 0, sizeof(unsigned long long) * 4 * drt_renderer::kRecords, r->stream));
            r->records_used = 0;
        }
        r->batch_first_record = r->records_used;
        for (uint32_t done = 0; done < n_frames; done += frames_per_launch) {
            fp.frame_first = r->frame_index + done;
            fp.n_frames = std::min(frames_per_launch, n_frames - done);
            unsigned long long *const record = r->spans_used < drt_renderer::kMaxSpans ? r->records + 4 * (size_t)(r->batch_first_record + r->spans_used++) : nullptr;
            if (record) r->records_used++;
            fp.span = record;
            unsigned int *const launch_status = record ? reinterpret_cast<unsigned int *>(record + 2) : nullptr;     // (beyond kMaxSpans launches per batch: no status word, the kernel still aborts cleanly)
            r->launches_last++;
            if (r->counters_used == drt_renderer::kCounters) {      // stream order: every launch that used them is over by then
                HIP_TRY(hipMemsetAsync(r->tile_counter, 0, sizeof(unsigned int) * drt_renderer::kCounters * kQueueHeadBlockWords, r->stream));
                r->counters_used = 0;
            }
            unsigned int *const queue_head = r->tile_counter + (size_t)(r->counters_used++) * kQueueHeadBlockWords;
            bool pool_hbm_scene = false;
            const bool use_pool = r->use_path_pool && path_pool_supports(r->view, fp, r->bvh_depth, wave_queue_scene_lds_bytes(r->view), &pool_hbm_scene);
            if (use_pool) r->pool_launched = true;
            if (use_pool)
                HIP_TRY(launch_path_pool(r->view, fp, r->bvh_depth, r->scene_has_alpha, pool_hbm_scene, r->pool_t_class, r->pool_tuning, r->pool_scratch, queue_head, r->samples, launch_status,
                                         r->num_cus, r->stream, &r->kernel_name, r->launch_shape));
            else if (fp.ext_transmission)
                return fail(DRT_ERR_UNSUPPORTED, "the dielectric lobe of drt_material_model is rendered by path_pool only (not: debug views, DRT_KERNEL=wave_queue, trees beyond 32 767 nodes, bounce limits beyond 30 000)");
            else
            HIP_TRY(launch_wave_queue(r->view, fp, r->bvh_depth, r->counting ? 2 : (material_ext ? 1 : 0), r->scene_has_alpha, queue_head,
                                      r->samples, r->num_cus, r->stream, &r->kernel_name, r->launch_shape, r->wq_cache));
        }
    }
    // the launches' records (execution span, status bits) travel to pinned host memory on the stream: drt_renderer_wait reads them
    // after the event, no second round trip to the device (a 1/8-shard step is 0.4 ms)
    if (r->spans_used > 0)
        HIP_TRY(hipMemcpyAsync(r->records_host, r->records + 4 * (size_t)r->batch_first_record, sizeof(unsigned long long) * 4 * (size_t)r->spans_used, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipEventRecord(r->ev_stop, r->stream));                    // Renderer.cu:105
    r->frame_index += n_frames;                                        // Renderer.cu:116
    r->pending = true;
    if (!blocking) return DRT_OK;
    return drt_renderer_wait(r, delta_ms);                             // blocking, Renderer.cu:106-108
}

int drt_renderer_render_batch(drt_renderer *r, const drt_camera *cam, const drt_scene *scene, uint32_t n_frames,
                              float *delta_ms) {
    return render_batch_impl(r, cam, scene, n_frames, delta_ms, true);
}

int drt_renderer_render_batch_async(drt_renderer *r, const drt_camera *cam, const drt_scene *scene, uint32_t n_frames) {
    return render_batch_impl(r, cam, scene, n_frames, nullptr, false);
}

int drt_renderer_wait(drt_renderer *r, float *delta_ms) {
    if (!r) return fail(DRT_ERR_INVALID, "null renderer");
    if (delta_ms) *delta_ms = 0.f;
    if (!r->pending) return DRT_OK;
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipEventSynchronize(r->ev_stop));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, r->ev_start, r->ev_stop));
    if (delta_ms) *delta_ms = ms;
    r->span_ms = 0.f;
    unsigned int status = 0;
    if (r->spans_used > 0) {
        const unsigned long long *host = r->records_host;          // (copied on the stream before ev_stop)
        double ticks = 0;
        for (int i = 0; i < r->spans_used; i++) {
            const unsigned long long start = ~host[4 * i], end = host[4 * i + 1];
            if (host[4 * i] != 0 && end > start) ticks += (double)(end - start);
            status |= (unsigned int)host[4 * i + 2];
        }
        r->span_ms = (float)(ticks / (double)r->wall_clock_khz);
    }
    wave_queue_report(r->wq_cache, r->span_ms);
    r->pending = false;
    if (r->pool_launched) {
        r->pool_launched = false;
        if (status != 0)
            return fail(DRT_ERR_DEVICE, "path_pool kernel: status " + std::to_string(status) + (status < 0x100u ? " (a queue wait exceeded its bound; the launch was abandoned)"
                                                                                                 : " (bits 8..: an index out of range was caught and clamped -- 0x100 triangle, 0x200 node, 0x400 leaf, "
                                                                                                   "0x800 / 0x1000 hit triangle, 0x2000 material, 0x4000 texture, 0x8000 sample slot, 0x10000 / 0x20000 stack level, "
                                                                                                   "0x40000 path id from a queue, 0x80000 shading record)"));
    }
    return DRT_OK;
}

int drt_renderer_render(drt_renderer *r, const drt_camera *cam, const drt_scene *scene, float *delta_ms) {
    return drt_renderer_render_batch(r, cam, scene, 1, delta_ms);
}

static int read_back(drt_renderer *r, const float *src, int comps, float *dst, size_t dst_floats) {
    if (!r || !dst) return fail(DRT_ERR_INVALID, "null argument");
    size_t need = (size_t)r->width * r->local_rows * (size_t)comps;
    if (dst_floats < need) return fail(DRT_ERR_INVALID, "destination too small");
    if (need == 0) return DRT_OK;
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipStreamSynchronize(r->stream));
    HIP_TRY(hipMemcpy(dst, src, need * sizeof(float), hipMemcpyDeviceToHost));
    return DRT_OK;
}

int drt_renderer_read_rgba32f(drt_renderer *r, float *dst, size_t dst_floats) {
    return read_back(r, r ? r->cur_rgba() : nullptr, 4, dst, dst_floats);
}
int drt_renderer_read_accum(drt_renderer *r, float *dst, size_t dst_floats) {
    return read_back(r, r ? r->cur_accum() : nullptr, 3, dst, dst_floats);
}

static int debug_check_exact(int32_t device, int which, uint64_t *mismatches, uint64_t *fast_path_count) {
    if (!mismatches || !fast_path_count) return fail(DRT_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(device));
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 2 * sizeof(unsigned long long)));
    hipError_t e = hipMemset(d, 0, 2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = launch_check_rcp(which, 0u, 1ull << 32, d, nullptr);
    unsigned long long h[2] = { 0, 0 };
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(DRT_ERR_DEVICE, hipGetErrorString(e));
    *mismatches = h[0]; *fast_path_count = h[1];
    return DRT_OK;
}

int drt_debug_check_rcp(int32_t device, uint64_t *mismatches, uint64_t *fast_path_count) {
    return debug_check_exact(device, 0, mismatches, fast_path_count);
}
int drt_debug_check_sqrt(int32_t device, uint64_t *mismatches, uint64_t *fast_path_count) {
    return debug_check_exact(device, 1, mismatches, fast_path_count);
}

int drt_debug_decode_image(const uint8_t *file, size_t file_bytes, drt_texture_info *info, uint8_t *out, size_t cap) {
    if (!file || !info) return fail(DRT_ERR_INVALID, "null argument");
    try {
        DecodedImage img;
        if (looks_like_png(file, file_bytes)) img = decode_png(file, file_bytes);
        else if (looks_like_jpeg(file, file_bytes)) img = decode_jpeg(file, file_bytes);
        else return fail(DRT_ERR_UNSUPPORTED, "neither PNG nor JPEG");
        info->width = img.width; info->height = img.height; info->components = img.components;
        if (out) {
            if (cap < img.texels.size()) return fail(DRT_ERR_INVALID, "destination too small");
            std::memcpy(out, img.texels.data(), img.texels.size());
        }
    } catch (...) { return from_exception(); }
    return DRT_OK;
}

int drt_debug_kat(int32_t device, int32_t which, const void *in, size_t in_bytes, void *out, size_t out_bytes, uint32_t n,
                  const drt_camera *cam, uint32_t width, uint32_t height) {
    if (!in || !out || which < 0 || which > 6) return fail(DRT_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    FrameParams fp;
    std::memset(&fp, 0, sizeof fp);
    if (which == 4) {
        if (!cam || width == 0 || height == 0) return fail(DRT_ERR_INVALID, "camera KAT needs a camera and a frame size");
        drt_renderer tmp;
        drt_default_settings(&tmp.settings);
        tmp.width = width; tmp.height = height;
        fill_frame_params(&tmp, cam, fp);
    }
    void *d_in = nullptr, *d_out = nullptr;
    HIP_TRY(hipMalloc(&d_in, std::max<size_t>(in_bytes, 16)));
    hipError_t e = hipMalloc(&d_out, std::max<size_t>(out_bytes, 16));
    if (e == hipSuccess) e = hipMemcpy(d_in, in, in_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_out, 0, out_bytes);
    if (e == hipSuccess) e = launch_kat(which, d_in, d_out, n, fp, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, out_bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return fail(DRT_ERR_DEVICE, hipGetErrorString(e));
    return DRT_OK;
}

int drt_debug_wave_queue_plans(const drt_renderer *r, char *buf, size_t cap) {
    if (!r || !buf || cap == 0) return fail(DRT_ERR_INVALID, "bad argument");
    std::string out = "[";
    for (const WqPlan &p : r->wq_cache.plans) {
        if (out.size() > 1) out += ", ";
        out += "{\"key\": " + std::to_string(p.key) + ", \"chosen\": " + std::to_string(p.chosen) + ", \"candidates\": [";
        for (size_t i = 0; i < p.cands.size(); i++) {
            char item[200];
            std::snprintf(item, sizeof item, "%s{\"threads\": %d, \"entry_bytes\": %d, \"tris\": %d, \"wg_per_cu\": %d, \"trials\": %d, \"ns_per_sample\": %.6f}",
                          i ? ", " : "", p.cands[i].threads, p.cands[i].entry_bytes, p.cands[i].tris, p.cands[i].per_cu, p.trials[i], p.ns_per_sample[i]);
            out += item;
        }
        out += "]}";
    }
    out += "]";
    if (out.size() + 1 > cap) return fail(DRT_ERR_INVALID, "destination too small");
    std::memcpy(buf, out.c_str(), out.size() + 1);
    return DRT_OK;
}

int drt_debug_pool_stats(drt_renderer *r, uint64_t out[40], int32_t reset) {
    if (!r || !out) return fail(DRT_ERR_INVALID, "null argument");
    if (!r->pool_tuning.stats) return fail(DRT_ERR_INVALID, "renderer was not created with DRT_POOL_STATS=1");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipStreamSynchronize(r->stream));
    HIP_TRY(hipMemcpy(out, r->pool_tuning.stats, 40 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(r->pool_tuning.stats, 0, 40 * sizeof(uint64_t)));
    return DRT_OK;
}

int drt_debug_hash_cycles(int32_t device, uint32_t max_len, uint32_t *pairs_out, uint32_t cap_pairs, uint32_t *found) {
    if (!pairs_out || !found || cap_pairs == 0) return fail(DRT_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    uint32_t *d = nullptr;
    const size_t bytes = (1 + 2 * (size_t)cap_pairs) * sizeof(uint32_t);
    HIP_TRY(hipMalloc((void **)&d, bytes));
    hipError_t e = hipMemset(d, 0, bytes);
    if (e == hipSuccess) e = launch_hash_cycles(max_len, d, cap_pairs, nullptr);
    std::vector<uint32_t> h(1 + 2 * (size_t)cap_pairs);
    if (e == hipSuccess) e = hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(DRT_ERR_DEVICE, hipGetErrorString(e));
    *found = h[0];
    std::memcpy(pairs_out, h.data() + 1, 2 * (size_t)std::min<uint32_t>(h[0], cap_pairs) * sizeof(uint32_t));
    return DRT_OK;
}

int drt_assemble_shards(const void *gathered, void *image, uint32_t width, uint32_t height, uint32_t stripe_rows,
                        uint32_t world, uint32_t padded_rows, void *hip_stream) {
    if (!gathered || !image || stripe_rows == 0 || world == 0) return fail(DRT_ERR_INVALID, "bad argument");
    if (padded_rows < drt_shard_rows(height, stripe_rows, 0, world)) return fail(DRT_ERR_INVALID, "padded_rows smaller than rank 0's shard");
    (void)hipGetLastError();                 // (see render_batch_impl: only this launch's own error counts)
    HIP_TRY(launch_assemble(gathered, image, width, height, stripe_rows, world, padded_rows, (hipStream_t)hip_stream));
    return DRT_OK;
}

}  // extern "C"
