// ref_kat_driver.cpp -- known-answer-test driver around the REFERENCE's own leaf functions.
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; everything it calls is compiled
// straight from /root/reference/DustRayTracer/src (see oracle/Makefile, target
// _ref/ref_kat): Core/Bounds.cu, Core/CudaMath/Random.cu,
// Core/Kernel/Shaders/Intersection.cu, Core/Scene/Camera.cu,
// Core/Scene/Texture.cu, Core/Interval.cu, and the header-only
// Core/Kernel/Shaders/ClosestHit.cuh, Core/Kernel/Shaders/Miss.cuh, Core/Interval.cuh.  Headers come from the image: the
// CUDA toolkit headers bundled with triton, glm and stb_image vendored by the
// reference.  No stand-in headers or libraries are written; the few CUDA runtime
// symbols referenced by code paths we never call stay unresolved.
//
// usage: ref_kat <function> <in.bin> <out.bin>     (raw little-endian arrays)
#include "Core/Bounds.cuh"
#include "Core/Ray.cuh"
#include "Core/CudaMath/Random.cuh"
#include "Core/Kernel/Shaders/Intersection.cuh"
#include "Core/Scene/Camera.cuh"
#include "Core/Scene/Texture.cuh"
#include "Core/Scene/Triangle.cuh"
#include "Core/HitPayload.cuh"
#include "Core/Kernel/Shaders/ClosestHit.cuh"
#include "Core/Kernel/Shaders/Miss.cuh"
#include "Core/Interval.cuh"
#include "Core/BVH/BVHNode.cuh"
#include "Core/Scene/Material.cuh"
#include "Core/Scene/Mesh.cuh"
#include "Core/Scene/RendererSettings.h"
#include <cstddef>
#include "stb_image.h"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static std::vector<unsigned char> slurp(const char *path)
{
    std::vector<unsigned char> v;
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    v.resize((size_t)n);
    if (n && fread(v.data(), 1, (size_t)n, f) != (size_t)n) { perror("read"); exit(2); }
    fclose(f);
    return v;
}

struct Out {
    std::vector<unsigned char> b;
    template <class T> void put(const T &v) { const unsigned char *p = (const unsigned char *)&v; b.insert(b.end(), p, p + sizeof(T)); }
    void put3(float3 v) { put(v.x); put(v.y); put(v.z); }
};

// Same layout as Core/Scene/Texture.cuh:14-19 (d_data is private; the only
// constructors need the CUDA runtime, so the object is filled bytewise).
struct TextureBits { const char *name; int width, height; int componentCount; unsigned char *d_data; };
static_assert(sizeof(TextureBits) == sizeof(Texture), "Texture layout");

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: %s <function> <in.bin> <out.bin>\n", argv[0]); return 2; }
    std::string fn = argv[1];
    std::vector<unsigned char> in = slurp(argv[2]);
    Out out;
    const float *fin = (const float *)in.data();
    const uint32_t *uin = (const uint32_t *)in.data();

    if (fn == "pcg") {                                   // in: u32[n] -> out: u32[n]
        for (size_t i = 0; i < in.size() / 4; i++) out.put((uint32_t)pcg_hash(uin[i]));
    } else if (fn == "randfloat") {                      // in: u32 seed, u32 n -> out: f32[n], u32 seed
        uint32_t seed = uin[0], n = uin[1];
        for (uint32_t i = 0; i < n; i++) out.put(randomFloat(seed));
        out.put(seed);
    } else if (fn == "unitvec" || fn == "unitsphere") {  // in: u32[n] -> out: n x (f32[3], u32 seed)
        for (size_t i = 0; i < in.size() / 4; i++) {
            uint32_t seed = uin[i];
            float3 v = fn == "unitvec" ? randomUnitVec3(seed) : randomUnitSphereVec3(seed);
            out.put3(v); out.put(seed);
        }
    } else if (fn == "unitdisk") {                       // in: u32[n] -> out: n x (f32[2], u32 seed)
        for (size_t i = 0; i < in.size() / 4; i++) {
            uint32_t seed = uin[i];
            float2 v = random_in_unit_disk(seed);
            out.put(v.x); out.put(v.y); out.put(seed);
        }
    } else if (fn == "slab") {                           // in: n x (orig3, dir3, min3, max3) -> out: f32[n]
        for (size_t i = 0; i + 12 <= in.size() / 4; i += 12) {
            Ray r(make_float3(fin[i], fin[i + 1], fin[i + 2]), make_float3(fin[i + 3], fin[i + 4], fin[i + 5]));
            Bounds3f b(make_float3(fin[i + 6], fin[i + 7], fin[i + 8]), make_float3(fin[i + 9], fin[i + 10], fin[i + 11]));
            out.put(b.intersect(r));
        }
    } else if (fn == "intersect") {                      // in: n x (orig3, dir3, v0, v1, v2) -> out: n x (t, U, V, W, i32 hit)
        for (size_t i = 0; i + 15 <= in.size() / 4; i += 15) {
            Ray r(make_float3(fin[i], fin[i + 1], fin[i + 2]), make_float3(fin[i + 3], fin[i + 4], fin[i + 5]));
            Triangle t;
            t.vertex0.position = make_float3(fin[i + 6], fin[i + 7], fin[i + 8]);
            t.vertex1.position = make_float3(fin[i + 9], fin[i + 10], fin[i + 11]);
            t.vertex2.position = make_float3(fin[i + 12], fin[i + 13], fin[i + 14]);
            ShortHitPayload p = Intersection(r, &t);
            int32_t hit = p.primitiveptr != nullptr;
            out.put(p.hit_distance);
            // UVW is left uninitialised by the reference on a miss
            out.put(hit ? p.UVW.x : 0.f); out.put(hit ? p.UVW.y : 0.f); out.put(hit ? p.UVW.z : 0.f);
            out.put(hit);
        }
    } else if (fn == "closesthit") {                     // in: n x (orig3, dir3, t, face_normal3) -> out: n x (pos3, normal3, i32 front_face)
        for (size_t i = 0; i + 10 <= in.size() / 4; i += 10) {
            Ray r(make_float3(fin[i], fin[i + 1], fin[i + 2]), make_float3(fin[i + 3], fin[i + 4], fin[i + 5]));
            Triangle t;
            t.face_normal = make_float3(fin[i + 7], fin[i + 8], fin[i + 9]);
            HitPayload hit;
            hit.hit_distance = fin[i + 6];
            hit.primitiveptr = &t;
            hit.UVW = make_float3(0, 0, 0);
            HitPayload p = ClosestHit(r, &hit);
            out.put3(p.world_position); out.put3(p.world_normal); out.put((int32_t)p.front_face);
        }
    } else if (fn == "cammove") {
        // Camera::Rotate / Camera::OnUpdate (host logic, Camera.cu:44-80) applied in sequence to one camera.
        // in: pos3, fwd3, up3, right3, speed, then n x (sin_x, cos_x, sin_y, cos_y, vel3, delta) -> out: n x (pos3, fwd3, right3)
        Camera cam(make_float3(fin[0], fin[1], fin[2]));
        cam.m_Forward_dir = make_float3(fin[3], fin[4], fin[5]);
        cam.m_Up_dir = make_float3(fin[6], fin[7], fin[8]);
        cam.m_Right_dir = make_float3(fin[9], fin[10], fin[11]);
        cam.setMovementSpeed(fin[12]);
        for (size_t i = 13; i + 8 <= in.size() / 4; i += 8) {
            cam.OnUpdate(make_float3(fin[i + 4], fin[i + 5], fin[i + 6]), fin[i + 7]);          // EditorLayer.cpp:400-401 order
            cam.Rotate(make_float4(fin[i], fin[i + 1], fin[i + 2], fin[i + 3]));
            out.put3(cam.GetPosition()); out.put3(cam.m_Forward_dir); out.put3(cam.m_Right_dir);
        }
    } else if (fn == "layout") {
        // sizes / field offsets of the structs the C ABI mirrors, and the default member values of settings and camera
        // (as "key=value" lines; floats as their bit patterns)
        auto kv = [&](const char *k, long long v) { char b[128]; int n = snprintf(b, sizeof b, "%s=%lld\n", k, v); out.b.insert(out.b.end(), b, b + n); };
        auto kf = [&](const char *k, float v) { uint32_t u; memcpy(&u, &v, 4); kv(k, (long long)u); };
        kv("sizeof_Vertex", sizeof(Vertex)); kv("Vertex.position", offsetof(Vertex, position)); kv("Vertex.normal", offsetof(Vertex, normal)); kv("Vertex.UV", offsetof(Vertex, UV));
        kv("sizeof_Triangle", sizeof(Triangle)); kv("Triangle.centroid", offsetof(Triangle, centroid)); kv("Triangle.vertex0", offsetof(Triangle, vertex0));
        kv("Triangle.vertex1", offsetof(Triangle, vertex1)); kv("Triangle.vertex2", offsetof(Triangle, vertex2));
        kv("Triangle.face_normal", offsetof(Triangle, face_normal)); kv("Triangle.materialIdx", offsetof(Triangle, materialIdx));
        kv("sizeof_BVHNode", sizeof(BVHNode)); kv("BVHNode.m_IsLeaf", offsetof(BVHNode, m_IsLeaf)); kv("BVHNode.m_BoundingBox", offsetof(BVHNode, m_BoundingBox));
        kv("BVHNode.dev_child1_idx", offsetof(BVHNode, dev_child1_idx)); kv("BVHNode.dev_child2_idx", offsetof(BVHNode, dev_child2_idx));
        kv("BVHNode.primitives_count", offsetof(BVHNode, primitives_count)); kv("BVHNode.primitive_start_idx", offsetof(BVHNode, primitive_start_idx));
        kv("BVHNode.rayint_cost", BVHNode::rayint_cost); kv("BVHNode.trav_cost", BVHNode::trav_cost);
        kv("sizeof_Material", sizeof(Material)); kv("Material.Albedo", offsetof(Material, Albedo)); kv("Material.EmmisiveFactor", offsetof(Material, EmmisiveFactor));
        kv("Material.AlbedoTextureIndex", offsetof(Material, AlbedoTextureIndex)); kv("Material.Roughness", offsetof(Material, Roughness));
        kv("Material.Transmission", offsetof(Material, Transmission)); kv("Material.refractive_index", offsetof(Material, refractive_index));
        kv("Material.Metallic", offsetof(Material, Metallic));
        Material m;
        kf("Material.default.Albedo.x", m.Albedo.x); kv("Material.default.AlbedoTextureIndex", m.AlbedoTextureIndex); kf("Material.default.refractive_index", m.refractive_index);
        RendererSettings rs;
        kv("settings.gamma_correction", rs.gamma_correction); kv("settings.tone_mapping", rs.tone_mapping); kv("settings.enableSunlight", rs.enableSunlight);
        kv("settings.max_samples", rs.max_samples); kv("settings.ray_bounce_limit", rs.ray_bounce_limit);
        kv("settings.RenderMode", (int)rs.RenderMode); kv("settings.DebugMode", (int)rs.DebugMode);
        kf("settings.sunlight_dir.x", rs.sunlight_dir.x); kf("settings.sunlight_dir.y", rs.sunlight_dir.y);
        kf("settings.sunlight_color.x", rs.sunlight_color.x); kf("settings.sunlight_color.y", rs.sunlight_color.y); kf("settings.sunlight_color.z", rs.sunlight_color.z);
        kf("settings.sunlight_intensity", rs.sunlight_intensity);
        kf("settings.sky_color.x", rs.sky_color.x); kf("settings.sky_color.y", rs.sky_color.y); kf("settings.sky_color.z", rs.sky_color.z);
        kf("settings.sky_intensity", rs.sky_intensity);
        Camera cam;
        kf("camera.exposure", cam.exposure); kf("camera.vfov_rad", cam.vfov_rad); kf("camera.defocus_angle", cam.defocus_angle);
        kf("camera.focus_dist", cam.focus_dist); kf("camera.m_movement_speed", cam.m_movement_speed);
        kf("camera.m_Position.x", cam.m_Position.x); kf("camera.m_Position.y", cam.m_Position.y); kf("camera.m_Position.z", cam.m_Position.z);
        kf("camera.m_Forward_dir.x", cam.m_Forward_dir.x); kf("camera.m_Forward_dir.y", cam.m_Forward_dir.y); kf("camera.m_Forward_dir.z", cam.m_Forward_dir.z);
        kf("camera.m_Up_dir.x", cam.m_Up_dir.x); kf("camera.m_Up_dir.y", cam.m_Up_dir.y); kf("camera.m_Up_dir.z", cam.m_Up_dir.z);
        kf("camera.m_Right_dir.x", cam.m_Right_dir.x); kf("camera.m_Right_dir.y", cam.m_Right_dir.y); kf("camera.m_Right_dir.z", cam.m_Right_dir.z);
        kf("deg2rad_60", deg2rad(60));
    } else if (fn == "centroid") {                       // Triangle.cuh:9-12; in: n x (p0, p1, p2) -> out: n x centroid3
        for (size_t i = 0; i + 9 <= in.size() / 4; i += 9) {
            Vertex v0(make_float3(fin[i], fin[i + 1], fin[i + 2]), make_float3(0, 0, 0), make_float2(0, 0));
            Vertex v1(make_float3(fin[i + 3], fin[i + 4], fin[i + 5]), make_float3(0, 0, 0), make_float2(0, 0));
            Vertex v2(make_float3(fin[i + 6], fin[i + 7], fin[i + 8]), make_float3(0, 0, 0), make_float2(0, 0));
            Triangle t(v0, v1, v2, make_float3(0, 0, 1), 0);
            out.put3(t.centroid);
        }
    } else if (fn == "surfacearea") {                    // Bounds.cu:4-10 via BVHNode::getSurfaceArea (BVHNode.cuh:29-35); in: n x (min3, max3, i32 count) -> out: f32[n]
        for (size_t i = 0; i + 7 <= in.size() / 4; i += 7) {
            BVHNode node;
            node.m_BoundingBox = Bounds3f(make_float3(fin[i], fin[i + 1], fin[i + 2]), make_float3(fin[i + 3], fin[i + 4], fin[i + 5]));
            node.primitives_count = (int)uin[i + 6];
            out.put(node.getSurfaceArea());
        }
    } else if (fn == "surrounds") {                      // Interval::surrounds (Interval.cuh:24-26); in: n x (min, max, x) -> out: i32[n]
        for (size_t i = 0; i + 3 <= in.size() / 4; i += 3) {
            Interval iv(fin[i], fin[i + 1]);
            out.put((int32_t)iv.surrounds(fin[i + 2]));
        }
    } else if (fn == "miss") {                           // Miss (Shaders/Miss.cuh:2-6); in: n x (orig3, dir3, color3) -> out: n x (color3, hit_distance, i32 has_prim, i32 front_face)
        for (size_t i = 0; i + 9 <= in.size() / 4; i += 9) {
            Ray r(make_float3(fin[i], fin[i + 1], fin[i + 2]), make_float3(fin[i + 3], fin[i + 4], fin[i + 5]));
            HitPayload p = Miss(r, make_float3(fin[i + 6], fin[i + 7], fin[i + 8]));
            out.put3(p.color); out.put(p.hit_distance); out.put((int32_t)(p.primitiveptr != nullptr)); out.put((int32_t)p.front_face);
        }
    } else if (fn == "boundscentroid") {                 // Bounds3f::getCentroid (Bounds.cu:12-15); in: n x (min3, max3) -> out: n x centroid3
        for (size_t i = 0; i + 6 <= in.size() / 4; i += 6) {
            Bounds3f b(make_float3(fin[i], fin[i + 1], fin[i + 2]), make_float3(fin[i + 3], fin[i + 4], fin[i + 5]));
            out.put3(b.getCentroid());
        }
    } else if (fn == "stbload") {                        // Texture.cu:21-30: in: an image file's bytes -> out: i32 w, h, comps, then the texels
        int w = 0, h = 0, n = 0;
        unsigned char *px = stbi_load_from_memory(in.data(), (int)in.size(), &w, &h, &n, 0);
        out.put((int32_t)w); out.put((int32_t)h); out.put((int32_t)n);
        if (px) out.b.insert(out.b.end(), px, px + (size_t)w * h * n);
        else { out.b.clear(); out.put((int32_t)0); out.put((int32_t)0); out.put((int32_t)0); }
    } else if (fn == "getray") {
        // in: exposure, vfov_rad, defocus_angle, focus_dist, pos3, fwd3, width, height, then n x (u, v, u32 seed)
        Camera cam(make_float3(fin[4], fin[5], fin[6]));
        cam.exposure = fin[0]; cam.vfov_rad = fin[1]; cam.defocus_angle = fin[2]; cam.focus_dist = fin[3];
        cam.m_Forward_dir = make_float3(fin[7], fin[8], fin[9]);
        float width = fin[10], height = fin[11];
        for (size_t i = 12; i + 3 <= in.size() / 4; i += 3) {
            uint32_t seed = uin[i + 2];
            Ray r = cam.GetRay(make_float2(fin[i], fin[i + 1]), width, height, seed);
            out.put3(r.getOrigin()); out.put3(r.getDirection()); out.put(seed);
        }
    } else if (fn == "texpixel" || fn == "texalpha") {
        // in: i32 w, h, comps, n; n x (u, v); then texel bytes (padded by the caller)
        const int32_t *iin = (const int32_t *)in.data();
        int w = iin[0], h = iin[1], c = iin[2], n = iin[3];
        TextureBits bits = { "kat", w, h, c, in.data() + 16 + (size_t)n * 8 };
        Texture tex;
        memcpy((void *)&tex, &bits, sizeof bits);
        const float *uv = fin + 4;
        for (int i = 0; i < n; i++) {
            float2 q = make_float2(uv[2 * i], uv[2 * i + 1]);
            if (fn == "texpixel") out.put3(tex.getPixel(q)); else out.put(tex.getAlpha(q));
        }
    } else {
        fprintf(stderr, "unknown function %s\n", fn.c_str());
        return 2;
    }

    FILE *f = fopen(argv[3], "wb");
    if (!f) { perror(argv[3]); return 2; }
    if (!out.b.empty()) fwrite(out.b.data(), 1, out.b.size(), f);
    fclose(f);
    return 0;
}
