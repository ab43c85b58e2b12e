// editor_calls.cpp -- the statements Editor/EditorLayer.cpp makes against the core classes (lines cited), written against
// include/DustRayTracer.hpp: if this compiles, those editor lines compile unchanged.  Runs the scene / camera part on the
// CPU; with "--render" (needs a GPU) also the renderer part.
#include <DustRayTracer.hpp>

#include <cmath>
#include <cstdio>
#include <cstring>

struct DevMetrics { size_t m_ObjectsCount = 0, m_TrianglesCount = 0, m_MaterialsCount = 0, m_TexturesCount = 0; };

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s scene.glb [--render]\n", argv[0]); return 2; }
    const bool render = argc > 2 && std::strcmp(argv[2], "--render") == 0;
    try {
        // OnAttach, EditorLayer.cpp:35-67
        Camera *m_device_Camera = new Camera();
        m_device_Camera->m_movement_speed = 10.0;
        m_device_Camera->defocus_angle = 0.f;
        m_device_Camera->focus_dist = 10.f;
        Scene *m_Scene = new Scene();
        m_Scene->loadGLTFmodel(argv[1]);
        BVHBuilder bvhbuilder;
        bvhbuilder.m_TargetLeafPrimitivesCount = 20;
        bvhbuilder.m_BinCount = 8;
        m_Scene->d_BVHTreeRoot = bvhbuilder.buildIterative(m_Scene->m_PrimitivesBuffer, m_Scene->m_BVHNodes);
        DevMetrics m_DevMetrics;
        m_DevMetrics.m_ObjectsCount = m_Scene->m_Meshes.size();
        for (Mesh mesh : m_Scene->m_Meshes) m_DevMetrics.m_TrianglesCount += mesh.m_trisCount;
        m_DevMetrics.m_MaterialsCount = m_Scene->m_Material.size();
        m_DevMetrics.m_TexturesCount = m_Scene->m_Textures.size();
        std::printf("objects=%zu triangles=%zu materials=%zu textures=%zu root=%d\n", m_DevMetrics.m_ObjectsCount, m_DevMetrics.m_TrianglesCount,
                    m_DevMetrics.m_MaterialsCount, m_DevMetrics.m_TexturesCount, m_Scene->d_BVHTreeRoot != nullptr);

        // camera input, EditorLayer.cpp:388-401, and the position / direction read-outs of :218-226
        Camera *cam = m_device_Camera;
        const float rotX = 0.01f, rotY = -0.02f, delta = 0.016f;
        float3_ velocity = { 0, 0, 1 };
        float4_ mousedeltadegrees = { std::sin(-rotY), std::cos(-rotY), std::sin(-rotX), std::cos(-rotX) };
        cam->OnUpdate(velocity, delta);
        cam->Rotate(mousedeltadegrees);
        float3_ pos = m_device_Camera->GetPosition();
        float3_ fdir = m_device_Camera->m_Forward_dir;
        std::printf("pos=%.6f,%.6f,%.6f fwd=%.6f,%.6f,%.6f\n", pos.x, pos.y, pos.z, fdir.x, fdir.y, fdir.z);

        // the settings widgets take addresses of these members, EditorLayer.cpp:241-277
        RendererSettings settings;
        bool *b0 = &settings.enableSunlight, *b1 = &settings.gamma_correction, *b2 = &settings.tone_mapping;
        int *i0 = &settings.ray_bounce_limit, *i1 = &settings.max_samples;
        float *f0 = (float *)&settings.sunlight_color, *f1 = &settings.sunlight_intensity, *f2 = &settings.sunlight_dir.x,
              *f3 = &settings.sunlight_dir.y, *f4 = (float *)&settings.sky_color, *f5 = &settings.sky_intensity;
        float *c0 = &m_device_Camera->vfov_rad, *c1 = &m_device_Camera->focus_dist, *c2 = &m_device_Camera->defocus_angle, *c3 = &m_device_Camera->exposure;
        int renderer_mode = (int)settings.RenderMode, debug_view = (int)settings.DebugMode;
        settings.RenderMode = (RendererSettings::RenderModes)renderer_mode;
        settings.DebugMode = (RendererSettings::DebugModes)debug_view;
        std::printf("settings: %d %d %d %d %d %.3f %.1f %.3f %.3f %.3f %.1f | %.4f %.1f %.1f %.1f\n", *b0, *b1, *b2, *i0, *i1, f0[1], *f1, *f2, *f3, f4[2], *f5,
                    *c0, *c1, *c2, *c3);

        // Core/Sampler.cuh:5-13 through its interface: the draws of pixel (57, 0), sample index 5 of a 1920-wide frame = seed 285,
        // then the known-answer sequence of randomFloat from seed 12345 (tests/golden/kat_ref.npz, the reference's own Random.cu)
        PCGSampler pcg(1920);
        Sampler *sampler = &pcg;
        sampler->StartSampler(float2_{ 57.f, 0.f }, 5, 0);
        std::printf("sampler: seed=%u pixel=%.0f,%.0f\n", pcg.seed(), sampler->GetPixel2D().x, sampler->GetPixel2D().y);
        pcg.setSeed(12345);
        const float d0 = sampler->Get1DSample();
        const float2_ d12 = sampler->Get2DSample();
        std::printf("sampler: draws=%a,%a,%a\n", d0, d12.x, d12.y);

        if (render) {                              // OnUIRender / OnUpdate, EditorLayer.cpp:137,212,317-318,424
            Renderer m_Renderer;
            float m_LastRenderTime_ms = 0;
            m_Renderer.m_RendererSettings = settings;
            m_Renderer.ResizeBuffer(uint32_t(160.f), uint32_t(90.f));
            if (m_Renderer.getSampleCount() < (uint32_t)m_Renderer.m_RendererSettings.max_samples)
                m_Renderer.Render(m_device_Camera, (*m_Scene), &m_LastRenderTime_ms);
            m_Renderer.resetAccumulationBuffer();
            m_Renderer.Render(m_device_Camera, (*m_Scene), &m_LastRenderTime_ms);
            std::printf("rendered %u x %u, sample %u, %.3f ms\n", m_Renderer.getBufferWidth(), m_Renderer.getBufferHeight(), m_Renderer.getSampleCount(),
                        m_LastRenderTime_ms);
        }
        delete m_Scene;
        delete m_device_Camera;
    } catch (const drt::Error &e) {
        std::fprintf(stderr, "drt error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
