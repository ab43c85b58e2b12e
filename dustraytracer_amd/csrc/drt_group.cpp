// drt_group.cpp -- several GPUs of one node behind one object: the multi-GPU entry point of the C ABI.
//
// One process, N devices (SURVEY.md 5 "Distributed communication backend", 8(e)): a renderer per device renders its
// framebuffer stripes (drt_renderer_set_shard: 8-row stripes dealt round robin, RNG seeds from the global pixel index, so
// the assembled image is bit-identical to the one-GPU image), then the shards are gathered on device 0 over RCCL: ONE
// ncclSend / ncclRecv pair per peer and frame -- a peer's shard is contiguous in its own buffer -- into a staging buffer on
// device 0 ([rank][row of the shard]; device 0's renderer renders straight into its slot), each peer on its own xGMI link,
// then ONE kernel de-interleaves the stripes into the full image (drt_assemble_shards).  Nothing else is exchanged: the path
// has no data-path collective.  (Round 2 received every 8-row stripe at its rows of the image: no assemble pass, but 118
// send / receive pairs per 1080p frame on the critical path of a 0.4 ms step; DRT_GROUP_GATHER=stripes still does that, for
// timing the two against each other.)
// MORE THAN ONE DEVICE HAS NEVER RUN HERE (one-GPU boxes): what is tested is the address arithmetic (CPU), a group of one, and
// the RCCL plumbing with the one device sending to itself (DRT_GROUP_FORCE_RCCL).
// RCCL is loaded at run time (dlopen "librccl.so.1") when a group of more than one device is created: the library
// has no link-time dependency on it, and a process that already carries another copy (PyTorch ships its own) is not
// handed a second one unless it asks for a group.
#include "../../include/drt.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" int drt_internal_fail(int code, const char *msg);      // drt_capi.cpp: sets drt_last_error() for this thread

namespace {

constexpr uint32_t kStripeRows = 8;

// the handful of RCCL entry points used, bound by name
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *buf, size_t count, int datatype, int peer, void *comm, hipStream_t stream) = nullptr;
    int (*Recv)(void *buf, size_t count, int datatype, int peer, void *comm, hipStream_t stream) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    static constexpr int kFloat = 7;               // ncclFloat32 (rccl.h ncclDataType_t)
    std::string load() {
        if (lib) return "";
        for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return std::string("cannot load RCCL: ") + dlerror();
        auto sym = [&](const char *n) { return dlsym(lib, n); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv) return "RCCL library lacks an expected symbol";
        return "";
    }
};
Rccl g_rccl;

}  // namespace

struct drt_group {
    std::vector<int> devices;
    std::vector<drt_renderer *> renderers;
    std::vector<hipStream_t> streams;
    std::vector<void *> comms;                // RCCL communicators, one per device (empty for a group of one)
    float *image = nullptr;                   // device 0: the full RGBA32F frame, rows in place
    float *staging = nullptr;                 // device 0: [rank][padded_rows][width] RGBA32F -- the shards as the ranks hold them (slot 0 = device 0's own render target)
    float *accum0 = nullptr;                  // device 0: its renderer's accumulation buffer (bound together with staging slot 0)
    uint32_t padded_rows = 0;                 // rows of the largest shard (rank 0's)
    uint32_t width = 0, height = 0;
    bool pending = false;
    bool self_gather = false;                 // DRT_GROUP_FORCE_RCCL: device 0's own stripes go through RCCL too
    bool per_stripe = false;                  // DRT_GROUP_GATHER=stripes: round 2's gather (a send / receive pair per stripe, received in place)
    std::chrono::steady_clock::time_point t0;
};

#define GROUP_HIP(expr)                                                                                     \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return drt_internal_fail(DRT_ERR_DEVICE, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
    } while (0)
#define GROUP_NCCL(expr)                                                                                    \
    do {                                                                                                    \
        int e_ = (expr);                                                                                    \
        if (e_ != 0) return drt_internal_fail(DRT_ERR_DEVICE, (std::string(#expr) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e_) : "RCCL error")).c_str()); \
    } while (0)

extern "C" {

// Where stripe k of `rank` lives: in the rank's compact shard (source) and in the full image (destination), in floats of
// an RGBA32F frame.  Returns 0 when the rank has no k-th stripe.  (Stripe s of the image = rows [s*stripe_rows, ...); rank r
// owns the stripes s = r, r + world, ...; only the image's last stripe can be short, so every earlier one is full.)
int drt_shard_stripe(uint32_t width, uint32_t height, uint32_t stripe_rows, uint32_t rank, uint32_t world, uint32_t k,
                     uint64_t *src_offset_floats, uint64_t *dst_offset_floats, uint64_t *count_floats) {
    if (stripe_rows == 0 || world == 0 || rank >= world) return 0;
    const uint64_t s = (uint64_t)k * world + rank, first_row = s * stripe_rows;
    if (first_row >= height) return 0;
    const uint64_t rows = std::min<uint64_t>(stripe_rows, height - first_row);
    if (src_offset_floats) *src_offset_floats = (uint64_t)k * stripe_rows * width * 4u;
    if (dst_offset_floats) *dst_offset_floats = first_row * width * 4u;
    if (count_floats) *count_floats = rows * width * 4u;
    return 1;
}

drt_group *drt_group_create(const int32_t *devices, int32_t n_devices) {
    if (!devices || n_devices < 1 || n_devices > 64) { drt_internal_fail(DRT_ERR_INVALID, "bad device list"); return nullptr; }
    for (int i = 0; i < n_devices; i++)
        for (int j = 0; j < i; j++)
            if (devices[i] == devices[j]) { drt_internal_fail(DRT_ERR_INVALID, "a device appears twice in the group"); return nullptr; }
    drt_group *g = new (std::nothrow) drt_group();
    if (!g) { drt_internal_fail(DRT_ERR_INVALID, "out of host memory"); return nullptr; }
    g->devices.assign(devices, devices + n_devices);
    for (int i = 0; i < n_devices; i++) {
        drt_renderer *r = drt_renderer_create(devices[i]);
        if (!r) { drt_group_destroy(g); return nullptr; }
        g->renderers.push_back(r);
        hipStream_t st = nullptr;
        if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
            drt_internal_fail(DRT_ERR_DEVICE, "cannot create a stream for a device of the group");
            drt_group_destroy(g);
            return nullptr;
        }
        g->streams.push_back(st);
        drt_renderer_set_stream(r, st);
        drt_renderer_set_shard(r, kStripeRows, (uint32_t)i, (uint32_t)n_devices);
    }
    // DRT_GROUP_FORCE_RCCL=1: a group of ONE device also creates its communicator and gathers its stripes through
    // ncclSend / ncclRecv to itself -- the RCCL plumbing can then be exercised on a one-GPU box
    g->self_gather = n_devices == 1 && std::getenv("DRT_GROUP_FORCE_RCCL") && std::atoi(std::getenv("DRT_GROUP_FORCE_RCCL")) != 0;
    g->per_stripe = std::getenv("DRT_GROUP_GATHER") && std::strcmp(std::getenv("DRT_GROUP_GATHER"), "stripes") == 0;
    if (n_devices > 1 || g->self_gather) {
        const std::string err = g_rccl.load();
        if (!err.empty()) { drt_internal_fail(DRT_ERR_DEVICE, err.c_str()); drt_group_destroy(g); return nullptr; }
        g->comms.assign((size_t)n_devices, nullptr);
        const int rc = g_rccl.CommInitAll(g->comms.data(), n_devices, g->devices.data());
        if (rc != 0) {
            g->comms.clear();
            drt_internal_fail(DRT_ERR_DEVICE, (std::string("ncclCommInitAll: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error")).c_str());
            drt_group_destroy(g);
            return nullptr;
        }
    }
    return g;
}

void drt_group_destroy(drt_group *g) {
    if (!g) return;
    if (g->pending) (void)drt_group_wait(g, nullptr);      // sends / receives may still target what is freed below
    for (size_t i = 0; i < g->streams.size(); i++) if (hipSetDevice(g->devices[i]) == hipSuccess) (void)hipStreamSynchronize(g->streams[i]);
    for (void *c : g->comms) if (c && g_rccl.CommDestroy) g_rccl.CommDestroy(c);
    for (size_t i = 0; i < g->renderers.size(); i++) drt_renderer_destroy(g->renderers[i]);
    for (size_t i = 0; i < g->streams.size(); i++) { (void)hipSetDevice(g->devices[i]); (void)hipStreamDestroy(g->streams[i]); }
    if (!g->devices.empty()) (void)hipSetDevice(g->devices[0]);
    if (g->image) (void)hipFree(g->image);
    if (g->staging) (void)hipFree(g->staging);
    if (g->accum0) (void)hipFree(g->accum0);
    delete g;
}

int32_t drt_group_size(const drt_group *g) { return g ? (int32_t)g->renderers.size() : 0; }
drt_renderer *drt_group_renderer(drt_group *g, int32_t index) {
    return g && index >= 0 && (size_t)index < g->renderers.size() ? g->renderers[(size_t)index] : nullptr;
}
uint32_t drt_group_sample_count(const drt_group *g) { return g && !g->renderers.empty() ? drt_renderer_sample_count(g->renderers[0]) : 0; }
void *drt_group_device_rgba(drt_group *g) { return g ? g->image : nullptr; }

int drt_group_resize(drt_group *g, uint32_t width, uint32_t height) {
    if (!g) return drt_internal_fail(DRT_ERR_INVALID, "null group");
    if (width == g->width && height == g->height && g->image) return DRT_OK;          // Renderer.cu:29-31: same size, nothing happens
    if (g->pending) { const int rc = drt_group_wait(g, nullptr); if (rc != DRT_OK) return rc; }     // nothing in flight may target what is reallocated
    GROUP_HIP(hipSetDevice(g->devices[0]));
    { const int rc = drt_renderer_bind_buffers(g->renderers[0], nullptr, nullptr); if (rc != DRT_OK) return rc; }
    for (drt_renderer *r : g->renderers) {
        const int rc = drt_renderer_resize(r, width, height);
        if (rc != DRT_OK) return rc;
    }
    GROUP_HIP(hipSetDevice(g->devices[0]));
    if (g->image) { (void)hipFree(g->image); g->image = nullptr; }
    if (g->staging) { (void)hipFree(g->staging); g->staging = nullptr; }
    if (g->accum0) { (void)hipFree(g->accum0); g->accum0 = nullptr; }
    const uint32_t world = (uint32_t)g->renderers.size();
    g->padded_rows = drt_shard_rows(height, kStripeRows, 0, world);
    const size_t image_floats = std::max<size_t>((size_t)width * height, 1) * 4;
    const size_t shard_floats = std::max<size_t>((size_t)width * g->padded_rows, 1) * 4;
    GROUP_HIP(hipMalloc((void **)&g->image, image_floats * sizeof(float)));
    GROUP_HIP(hipMemset(g->image, 0, image_floats * sizeof(float)));
    const size_t slots = world + (g->self_gather ? 1u : 0u);           // (self-gather: one more slot, for the one device to receive its own shard in)
    GROUP_HIP(hipMalloc((void **)&g->staging, shard_floats * slots * sizeof(float)));
    GROUP_HIP(hipMemset(g->staging, 0, shard_floats * slots * sizeof(float)));
    GROUP_HIP(hipMalloc((void **)&g->accum0, shard_floats / 4 * 3 * sizeof(float)));
    g->width = width; g->height = height;
    // device 0 renders straight into slot 0 of the staging buffer: its shard needs no copy before the assemble pass
    { const int rc = drt_renderer_bind_buffers(g->renderers[0], g->accum0, g->staging); if (rc != DRT_OK) return rc; }
    return drt_renderer_reset(g->renderers[0]);
}

int drt_group_set_settings(drt_group *g, const drt_settings *s) {
    if (!g || !s) return drt_internal_fail(DRT_ERR_INVALID, "null argument");
    for (drt_renderer *r : g->renderers) { const int rc = drt_renderer_set_settings(r, s); if (rc != DRT_OK) return rc; }
    return DRT_OK;
}

int drt_group_reset(drt_group *g) {
    if (!g) return drt_internal_fail(DRT_ERR_INVALID, "null group");
    for (drt_renderer *r : g->renderers) { const int rc = drt_renderer_reset(r); if (rc != DRT_OK) return rc; }
    return DRT_OK;
}

// Renders frames f .. f+n-1 on every device (its stripes), then gathers the stripes into device 0's image.  Returns at
// once; drt_group_wait blocks until the image is complete.
int drt_group_render_batch_async(drt_group *g, const drt_camera *cam, const drt_scene *scene, uint32_t n_frames) {
    if (!g || !cam || !scene) return drt_internal_fail(DRT_ERR_INVALID, "null argument");
    if (g->width == 0 || g->height == 0) return drt_internal_fail(DRT_ERR_INVALID, "drt_group_resize has not been called");
    const uint32_t world = (uint32_t)g->renderers.size();
    g->t0 = std::chrono::steady_clock::now();
    // From here on work may be in flight on any device: whatever goes wrong below, the group is drained before the error is
    // returned (drt_group_wait; an open RCCL group is closed first), so that the caller may destroy or resize it safely.
    g->pending = true;
    int rc = DRT_OK;
    bool in_rccl_group = false;
    auto hip_ok = [&](hipError_t e, const char *what) { if (e != hipSuccess && rc == DRT_OK) rc = drt_internal_fail(DRT_ERR_DEVICE, (std::string(what) + ": " + hipGetErrorString(e)).c_str()); return e == hipSuccess; };
    auto nccl_ok = [&](int e, const char *what) { if (e != 0 && rc == DRT_OK) rc = drt_internal_fail(DRT_ERR_DEVICE, (std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "RCCL error")).c_str()); return e == 0; };
    for (drt_renderer *r : g->renderers) {                 // every device starts tracing before anything is gathered
        rc = drt_renderer_render_batch_async(r, cam, scene, n_frames);
        if (rc != DRT_OK) break;
    }
    const size_t shard_floats = (size_t)g->width * g->padded_rows * 4;
    if (rc == DRT_OK && g->per_stripe) {
        // ---- round 2's gather: every stripe received at its rows of the image ----
        // device 0's own stripes: one strided copy on its stream (behind its render): local stripe k -> image stripe k * world
        if (!g->self_gather && hip_ok(hipSetDevice(g->devices[0]), "hipSetDevice")) {
            const size_t stripe_bytes = (size_t)kStripeRows * g->width * 4 * sizeof(float);
            const uint32_t local_rows = drt_renderer_local_rows(g->renderers[0]);
            const uint32_t full = local_rows / kStripeRows, rest = local_rows % kStripeRows;
            const float *src = static_cast<const float *>(drt_renderer_device_rgba(g->renderers[0]));
            if (full) hip_ok(hipMemcpy2DAsync(g->image, stripe_bytes * world, src, stripe_bytes, stripe_bytes, full, hipMemcpyDeviceToDevice, g->streams[0]), "hipMemcpy2DAsync");
            uint64_t so, dof, cnt;
            if (rest && drt_shard_stripe(g->width, g->height, kStripeRows, 0, world, full, &so, &dof, &cnt))
                hip_ok(hipMemcpyAsync(g->image + dof, src + so, cnt * sizeof(float), hipMemcpyDeviceToDevice, g->streams[0]), "hipMemcpyAsync");
        }
        if (rc == DRT_OK && (world > 1 || g->self_gather) && nccl_ok(g_rccl.GroupStart(), "ncclGroupStart")) {
            in_rccl_group = true;
            for (uint32_t rank = g->self_gather ? 0 : 1; rank < world && rc == DRT_OK; rank++) {
                const float *src = static_cast<const float *>(drt_renderer_device_rgba(g->renderers[rank]));
                for (uint32_t k = 0; rc == DRT_OK; k++) {
                    uint64_t so, dof, cnt;
                    if (!drt_shard_stripe(g->width, g->height, kStripeRows, rank, world, k, &so, &dof, &cnt)) break;
                    if (nccl_ok(g_rccl.Send(src + so, (size_t)cnt, Rccl::kFloat, 0, g->comms[rank], g->streams[rank]), "ncclSend"))
                        nccl_ok(g_rccl.Recv(g->image + dof, (size_t)cnt, Rccl::kFloat, (int)rank, g->comms[0], g->streams[0]), "ncclRecv");
                }
            }
        }
    } else if (rc == DRT_OK) {
        // ---- one transfer per peer: its whole shard (contiguous where it was rendered) into its slot of the staging buffer ----
        if ((world > 1 || g->self_gather) && nccl_ok(g_rccl.GroupStart(), "ncclGroupStart")) {
            in_rccl_group = true;
            for (uint32_t rank = g->self_gather ? 0 : 1; rank < world && rc == DRT_OK; rank++) {
                const float *src = static_cast<const float *>(drt_renderer_device_rgba(g->renderers[rank]));
                const size_t count = (size_t)g->width * drt_renderer_local_rows(g->renderers[rank]) * 4;
                // (the self-gather of a one-device group: slot 0, where the device rendered, is sent to slot 1, and the assemble pass
                // below reads slot 1 -- the image then really is what RCCL moved)
                float *dst = g->staging + (size_t)(g->self_gather ? 1u : rank) * shard_floats;
                if (count == 0) continue;
                if (nccl_ok(g_rccl.Send(src, count, Rccl::kFloat, 0, g->comms[rank], g->streams[rank]), "ncclSend"))
                    nccl_ok(g_rccl.Recv(dst, count, Rccl::kFloat, (int)rank, g->comms[0], g->streams[0]), "ncclRecv");
            }
        }
    }
    if (in_rccl_group) nccl_ok(g_rccl.GroupEnd(), "ncclGroupEnd");          // (also on the error path: never leave the thread inside an open group)
    if (rc == DRT_OK && !g->per_stripe) {
        // the stripes of every shard to their rows of the image: one kernel on device 0's stream, behind its render and the receives
        if (hip_ok(hipSetDevice(g->devices[0]), "hipSetDevice"))
            rc = drt_assemble_shards(g->staging + (g->self_gather ? shard_floats : 0u), g->image, g->width, g->height, kStripeRows, world, g->padded_rows, g->streams[0]);
    }
    if (rc != DRT_OK) { (void)drt_group_wait(g, nullptr); return rc; }
    return DRT_OK;
}

int drt_group_wait(drt_group *g, float *delta_ms) {
    if (!g) return drt_internal_fail(DRT_ERR_INVALID, "null group");
    if (delta_ms) *delta_ms = 0.f;
    if (!g->pending) return DRT_OK;
    int first_error = DRT_OK;
    for (size_t i = 0; i < g->renderers.size(); i++) {
        const int rc = drt_renderer_wait(g->renderers[i], nullptr);
        if (rc != DRT_OK && first_error == DRT_OK) first_error = rc;
        if (hipSetDevice(g->devices[i]) != hipSuccess || hipStreamSynchronize(g->streams[i]) != hipSuccess)
            if (first_error == DRT_OK) first_error = drt_internal_fail(DRT_ERR_DEVICE, "stream synchronisation failed after the gather");
    }
    g->pending = false;
    if (delta_ms) *delta_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - g->t0).count();
    return first_error;
}

int drt_group_render_batch(drt_group *g, const drt_camera *cam, const drt_scene *scene, uint32_t n_frames, float *delta_ms) {
    const int rc = drt_group_render_batch_async(g, cam, scene, n_frames);
    if (rc != DRT_OK) return rc;
    return drt_group_wait(g, delta_ms);
}

int drt_group_read_rgba32f(drt_group *g, float *dst, size_t dst_floats) {
    if (!g || !dst) return drt_internal_fail(DRT_ERR_INVALID, "null argument");
    const size_t need = (size_t)g->width * g->height * 4;
    if (dst_floats < need) return drt_internal_fail(DRT_ERR_INVALID, "destination too small");
    if (g->pending) { const int rc = drt_group_wait(g, nullptr); if (rc != DRT_OK) return rc; }
    if (need == 0) return DRT_OK;
    GROUP_HIP(hipSetDevice(g->devices[0]));
    GROUP_HIP(hipMemcpy(dst, g->image, need * sizeof(float), hipMemcpyDeviceToHost));
    return DRT_OK;
}

}  // extern "C"
