// DustRayTracer.hpp -- the reference's "intended library header" (DustRayTracer/include/DustRayTracer.hpp:1 is empty)
// filled in for the MI355X core: thin C++ classes with the reference's names and members over the C ABI of drt.h.
//
//   Renderer        Core/Renderer.hpp:14-47          ResizeBuffer / Render / resetAccumulationBuffer / getSampleCount /
//                                                    getBufferWidth / getBufferHeight / public m_RendererSettings
//   Scene           Core/Scene/Scene.cuh:41-57       loadGLTFmodel (+ counts the editor shows, EditorLayer.cpp:57-65)
//   BVHBuilder      Core/BVH/BVHBuilder.cuh:12-23    m_BinCount, m_TargetLeafPrimitivesCount, buildIterative, build
//   Camera          Core/Scene/Camera.cuh:14-48      public fields, OnUpdate, Rotate, GetPosition
//   RendererSettings Core/Scene/RendererSettings.h   same fields and enums
//
// Differences a caller sees (INTEGRATION.md lists the editor-side edits):
//   * no GL interop: GetRenderTargetImage_name() is replaced by ReadRenderTarget(float*) / DeviceRenderTarget();
//   * Camera is a plain host object (no cudaMallocManaged `Managed` base), copied by value at each Render;
//   * errors throw drt::Error instead of printing and exit(99) (Editor/Common/CudaCommon.cu:4-13).
#pragma once
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "drt.h"

namespace drt {
struct Error : std::runtime_error {
    int code;
    Error(int c, const char *msg) : std::runtime_error(msg ? msg : "drt error"), code(c) {}
};
inline void check(int rc) { if (rc < 0) throw Error(rc, drt_last_error()); }
}  // namespace drt

struct float2_ { float x, y; };         // stand-ins for CUDA's vector types in this header's public fields
struct float3_ { float x, y, z; };
struct float4_ { float x, y, z, w; };

// Core/Scene/RendererSettings.h:4-35
struct RendererSettings {
    enum class RenderModes { NORMALMODE = 0, DEBUGMODE = 1 };
    enum class DebugModes { ALBEDO_DEBUG = 0, NORMAL_DEBUG = 1, BARYCENTRIC_DEBUG = 2, UVS_DEBUG = 3, MESHBVH_DEBUG = 4, WORLDBVH_DEBUG = 5 };
    bool gamma_correction = true;
    bool tone_mapping = true;
    bool enableSunlight = false;
    int max_samples = 500;
    int ray_bounce_limit = 2;
    RenderModes RenderMode = RenderModes::NORMALMODE;
    DebugModes DebugMode = DebugModes::ALBEDO_DEBUG;
    float2_ sunlight_dir = { -0.803f, 0.681f };
    float3_ sunlight_color = { 1.000f, 0.944f, 0.917f };
    float sunlight_intensity = 30;
    float3_ sky_color = { 0.25f, 0.498f, 0.80f };
    float sky_intensity = 20;

    drt_settings pod() const {
        drt_settings s;
        s.gamma_correction = gamma_correction; s.tone_mapping = tone_mapping; s.enable_sunlight = enableSunlight;
        s.max_samples = max_samples; s.ray_bounce_limit = ray_bounce_limit;
        s.render_mode = (int)RenderMode; s.debug_mode = (int)DebugMode;
        s.sunlight_dir[0] = sunlight_dir.x; s.sunlight_dir[1] = sunlight_dir.y;
        s.sunlight_color[0] = sunlight_color.x; s.sunlight_color[1] = sunlight_color.y; s.sunlight_color[2] = sunlight_color.z;
        s.sunlight_intensity = sunlight_intensity;
        s.sky_color[0] = sky_color.x; s.sky_color[1] = sky_color.y; s.sky_color[2] = sky_color.z;
        s.sky_intensity = sky_intensity;
        return s;
    }
};

inline float deg2rad(float degree) { const float PI = 3.14159265359f; return degree * (PI / 180.f); }    // Camera.cu:125-129

// Core/Scene/Camera.cuh:14-48
class Camera {
public:
    explicit Camera(float3_ pos = { 0, 2, 5 }) : m_Position(pos) { m_Right_dir = cross(m_Forward_dir, m_Up_dir); }

    void OnUpdate(float3_ velocity, float delta) {                       // Camera.cu:44-58
        drt_camera_move(&m_Position.x, &m_Right_dir.x, &m_Up_dir.x, &m_Forward_dir.x, &velocity.x, m_movement_speed, delta);
    }
    void Rotate(float4_ d) {                                             // Camera.cu:61-80 (sin_x, cos_x, sin_y, cos_y)
        drt_camera_rotate(&m_Forward_dir.x, &m_Right_dir.x, &m_Up_dir.x, &d.x);
    }
    float3_ GetPosition() const { return m_Position; }
    void setMovementSpeed(float speed) { m_movement_speed = speed; }

    float exposure = 1;
    float vfov_rad = deg2rad(60);
    float zfar = 0, znear = 0, m_AspectRatio = 0;
    float defocus_angle = 0;
    float focus_dist = 10;
    float m_movement_speed = 10;
    float3_ m_Position = { 0, 2, 5 };
    float3_ m_Forward_dir = { 0, 0, -1 };
    float3_ m_Up_dir = { 0, 1, 0 };
    float3_ m_Right_dir = { 0, 1, 0 };

    drt_camera pod() const {
        drt_camera c;
        c.exposure = exposure; c.vfov_rad = vfov_rad; c.defocus_angle = defocus_angle; c.focus_dist = focus_dist;
        c.position[0] = m_Position.x; c.position[1] = m_Position.y; c.position[2] = m_Position.z;
        c.forward[0] = m_Forward_dir.x; c.forward[1] = m_Forward_dir.y; c.forward[2] = m_Forward_dir.z;
        return c;
    }

private:
    static float3_ cross(float3_ a, float3_ b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
};

// Core/Scene/Mesh.cuh:8-18 (what the editor's metrics loop reads, EditorLayer.cpp:61-64)
struct Mesh { const char *name = ""; int m_primitives_offset = -1; size_t m_trisCount = 0; };

// Core/Scene/Scene.cuh:41-57
struct Scene {
    Scene() : m_PrimitivesBuffer{ this }, m_BVHNodes{ this }, handle(drt_scene_create()) {
        if (!handle) throw drt::Error(DRT_ERR_INVALID, drt_last_error());
    }
    ~Scene() { drt_scene_destroy(handle); }
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;

    // strict = true: read the file as the glTF 2.0 specification defines it (drt.h DRT_LOAD_STRICT) instead of as Scene.cu does
    bool loadGLTFmodel(const char *filepath, bool strict = false) {
        drt::check(drt_scene_load_gltf_ex(handle, filepath, strict ? DRT_LOAD_STRICT : 0u));
        refresh();
        return true;
    }

    // host-side copies of what the reference keeps in thrust device vectors; enough for `.size()` and a range-for
    std::vector<Mesh> m_Meshes;
    std::vector<drt_material> m_Material;
    std::vector<drt_texture_info> m_Textures;
    // m_PrimitivesBuffer / m_BVHNodes live inside the scene handle; these members only name them, so that
    // `scene.d_BVHTreeRoot = builder.buildIterative(scene.m_PrimitivesBuffer, scene.m_BVHNodes)` (EditorLayer.cpp:55) compiles
    struct Part { Scene *scene; };
    Part m_PrimitivesBuffer, m_BVHNodes;
    const void *d_BVHTreeRoot = nullptr;              // non-null once a BVH has been built

    size_t meshCount() const { return (size_t)drt_scene_mesh_count(handle); }
    size_t trianglesCount() const { return (size_t)drt_scene_triangle_count(handle); }
    size_t materialsCount() const { return (size_t)drt_scene_material_count(handle); }
    size_t texturesCount() const { return (size_t)drt_scene_texture_count(handle); }
    std::vector<drt_triangle> primitives() const {                        // m_PrimitivesBuffer
        std::vector<drt_triangle> v(trianglesCount());
        if (!v.empty()) drt::check(drt_scene_get_triangles(handle, v.data(), (int32_t)v.size()));
        return v;
    }
    std::vector<drt_bvh_node> bvhNodes() const {                          // m_BVHNodes (root last)
        std::vector<drt_bvh_node> v((size_t)drt_scene_node_count(handle));
        if (!v.empty()) drt::check(drt_scene_get_nodes(handle, v.data(), (int32_t)v.size()));
        return v;
    }
    void refresh() {                                                      // after anything that changes the scene
        std::vector<drt_mesh> meshes(meshCount());
        if (!meshes.empty()) drt::check(drt_scene_get_meshes(handle, meshes.data(), (int32_t)meshes.size()));
        m_Meshes.clear();
        for (const drt_mesh &m : meshes) { Mesh out; out.m_primitives_offset = m.primitives_offset; out.m_trisCount = (size_t)m.tris_count; m_Meshes.push_back(out); }
        m_Material.resize(materialsCount());
        if (!m_Material.empty()) drt::check(drt_scene_get_materials(handle, m_Material.data(), (int32_t)m_Material.size()));
        m_Textures.resize(texturesCount());
        for (size_t i = 0; i < m_Textures.size(); i++) drt::check(drt_scene_get_texture_info(handle, (int32_t)i, &m_Textures[i]));
    }

    drt_scene *handle;
};

// Core/BVH/BVHBuilder.cuh:12-23
class BVHBuilder {
public:
    int m_BinCount = 8;
    int m_TargetLeafPrimitivesCount = 6;
    int m_BuildDevice = -1;               // new: >= 0 runs the same build on that GPU (drt_scene_build_bvh_device)
    float m_LastBuildDeviceMs = 0;
    // the reference passes (scene.m_PrimitivesBuffer, scene.m_BVHNodes) and stores the returned root pointer
    // (EditorLayer.cpp:55); here the scene owns both, so the scene is the argument.
    void buildIterative(Scene &scene) {
        if (m_BuildDevice >= 0)
            drt::check(drt_scene_build_bvh_device(scene.handle, m_TargetLeafPrimitivesCount, m_BinCount, m_BuildDevice, &m_LastBuildDeviceMs));
        else
            drt::check(drt_scene_build_bvh(scene.handle, m_TargetLeafPrimitivesCount, m_BinCount));
        scene.d_BVHTreeRoot = scene.handle;
    }
    // BVHBuilder.cu:100-173: the same tree through recursion -- same nodes, same triangle order, node ARRAY in the recursion's
    // order (children after both of their subtrees); always on the host
    void build(Scene &scene) {
        drt::check(drt_scene_build_bvh_recursive(scene.handle, m_TargetLeafPrimitivesCount, m_BinCount));
        scene.d_BVHTreeRoot = scene.handle;
    }
    // the reference's spelling (EditorLayer.cpp:55): both arguments name parts of one scene; returns the root token
    const void *buildIterative(Scene::Part &primitives, Scene::Part &nodes) {
        if (primitives.scene != nodes.scene) throw drt::Error(DRT_ERR_INVALID, "primitives and nodes of different scenes");
        buildIterative(*primitives.scene);
        return primitives.scene->d_BVHTreeRoot;
    }
    const void *build(Scene::Part &primitives, Scene::Part &nodes) {
        if (primitives.scene != nodes.scene) throw drt::Error(DRT_ERR_INVALID, "primitives and nodes of different scenes");
        build(*primitives.scene);
        return primitives.scene->d_BVHTreeRoot;
    }
};

// Core/Sampler.cuh:5-13: the interface the reference declares and never implements (PCGSampler is an empty class there, RayGen.cuh
// seeds by hand).  Here PCGSampler is the RayGen recipe behind that interface, on the host: seed = (x + y * width) * sampleidx
// (RayGen.cuh:74-75), + dimension (the bounce index, :91), draws = randomFloat (Random.cu:13-17) -- the numbers a path of the
// kernels draws, for tools that want to follow one.
class Sampler {
public:
    virtual ~Sampler() = default;
    virtual void StartSampler(float2_ pixel, uint32_t sampleidx, int dimension) = 0;   // dim replaces bounces
    virtual float Get1DSample() = 0;
    virtual float2_ Get2DSample() = 0;
    virtual float2_ GetPixel2D() = 0;
};
class PCGSampler : public Sampler {
public:
    explicit PCGSampler(uint32_t image_width) : m_width(image_width) {}
    void StartSampler(float2_ pixel, uint32_t sampleidx, int dimension) override {
        m_pixel = pixel;
        m_seed = ((uint32_t)pixel.x + (uint32_t)pixel.y * m_width) * sampleidx + (uint32_t)dimension;
    }
    float Get1DSample() override { return drt_random_float(&m_seed); }
    float2_ Get2DSample() override { float2_ v; v.x = drt_random_float(&m_seed); v.y = drt_random_float(&m_seed); return v; }
    float2_ GetPixel2D() override { return m_pixel; }
    uint32_t seed() const { return m_seed; }
    void setSeed(uint32_t s) { m_seed = s; }
private:
    uint32_t m_width, m_seed = 0;
    float2_ m_pixel = { 0, 0 };
};

// Core/Renderer.hpp:14-47.  Renderer(device) renders on one GPU; Renderer({0, 1, ..., 7}) renders on several GPUs of the
// node (drt_group_*: framebuffer stripes per device, gathered into the first device's image over RCCL) behind the same calls.
class Renderer {
public:
    explicit Renderer(int device = 0) : handle(drt_renderer_create(device)) { if (!handle) throw drt::Error(DRT_ERR_DEVICE, drt_last_error()); }
    explicit Renderer(const std::vector<int> &devices) {
        std::vector<int32_t> d(devices.begin(), devices.end());
        group = drt_group_create(d.data(), (int32_t)d.size());
        if (!group) throw drt::Error(DRT_ERR_DEVICE, drt_last_error());
        handle = drt_group_renderer(group, 0);
    }
    ~Renderer() { if (group) drt_group_destroy(group); else drt_renderer_destroy(handle); }
    Renderer(const Renderer &) = delete;
    Renderer &operator=(const Renderer &) = delete;

    void ResizeBuffer(uint32_t width, uint32_t height) {
        drt::check(group ? drt_group_resize(group, width, height) : drt_renderer_resize(handle, width, height));
    }
    void Render(Camera *cam, const Scene &scene, float *delta) { RenderBatch(cam, scene, 1, delta); }
    // spp batch: frames getSampleCount() .. +n-1 in one launch, same image as n Render() calls
    void RenderBatch(Camera *cam, const Scene &scene, uint32_t n_frames, float *delta) {
        drt_settings s = m_RendererSettings.pod();
        drt_camera c = cam->pod();
        if (group) {
            drt::check(drt_group_set_settings(group, &s));
            drt::check(drt_group_render_batch(group, &c, scene.handle, n_frames, delta));
        } else {
            drt::check(drt_renderer_set_settings(handle, &s));
            drt::check(drt_renderer_render_batch(handle, &c, scene.handle, n_frames, delta));
        }
    }
    uint32_t getBufferWidth() const { return drt_renderer_width(handle); }
    uint32_t getBufferHeight() const { return drt_renderer_height(handle); }
    uint32_t getSampleCount() const { return drt_renderer_sample_count(handle); }
    void resetAccumulationBuffer() { drt::check(group ? drt_group_reset(group) : drt_renderer_reset(handle)); }
    int deviceCount() const { return group ? (int)drt_group_size(group) : 1; }
    // opt-in material model (drt.h drt_material_model; off = the reference's image): emissive term, metallic lobe, dielectric lobe
    void setMaterialModel(bool emissive, bool specular, float emissive_scale = 1.0f, bool transmission = false) {
        drt_material_model m = { emissive ? 1 : 0, specular ? 1 : 0, emissive_scale, transmission ? 1 : 0 };
        const int n = deviceCount();
        for (int i = 0; i < n; i++) drt::check(drt_renderer_set_material_model(group ? drt_group_renderer(group, i) : handle, &m));
    }

    // replaces GLuint& GetRenderTargetImage_name(): RGBA32F, row 0 = bottom, width*height*4 floats
    void ReadRenderTarget(float *dst) {
        const size_t n = (size_t)getBufferWidth() * getBufferHeight() * 4;
        drt::check(group ? drt_group_read_rgba32f(group, dst, n) : drt_renderer_read_rgba32f(handle, dst, n));
    }
    void *DeviceRenderTarget() { return group ? drt_group_device_rgba(group) : drt_renderer_device_rgba(handle); }

    RendererSettings m_RendererSettings;
    drt_renderer *handle = nullptr;       // the (first) device's renderer
    drt_group *group = nullptr;           // set when the renderer spans several devices
};
