// drt_render.cpp -- headless use of the reference-shaped C++ API: scene.glb -> RGBA32F -> PFM file.
//   g++ -std=c++17 -Iinclude examples/drt_render.cpp -Ldustraytracer_amd -ldrt_hip -Wl,-rpath,$PWD/dustraytracer_amd -o drt_render
//   ./drt_render models/cornell_box.glb out.pfm 1920 1080 8 8  3.6 1.25 0  -1 0 0
#include <DustRayTracer.hpp>

#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 7) {
        std::fprintf(stderr, "usage: %s scene.glb out.pfm width height spp depth [px py pz fx fy fz]\n", argv[0]);
        return 2;
    }
    try {
        const uint32_t W = (uint32_t)std::atoi(argv[3]), H = (uint32_t)std::atoi(argv[4]), spp = (uint32_t)std::atoi(argv[5]);
        Scene scene;
        scene.loadGLTFmodel(argv[1]);
        BVHBuilder builder;                               // EditorLayer.cpp:52-55
        builder.m_TargetLeafPrimitivesCount = 20;
        builder.m_BinCount = 8;
        builder.buildIterative(scene);
        Camera cam;
        if (argc >= 13) {
            cam.m_Position = { (float)std::atof(argv[7]), (float)std::atof(argv[8]), (float)std::atof(argv[9]) };
            cam.m_Forward_dir = { (float)std::atof(argv[10]), (float)std::atof(argv[11]), (float)std::atof(argv[12]) };
        }
        Renderer renderer(0);
        renderer.m_RendererSettings.ray_bounce_limit = std::atoi(argv[6]);
        renderer.m_RendererSettings.max_samples = (int)spp + 1;
        renderer.ResizeBuffer(W, H);
        float ms = 0;
        renderer.RenderBatch(&cam, scene, spp, &ms);
        std::vector<float> rgba((size_t)W * H * 4);
        renderer.ReadRenderTarget(rgba.data());
        std::printf("%zu triangles, %u x %u, %u spp: %.3f ms (%.1f Msamples/s)\n", scene.trianglesCount(), W, H, spp, ms,
                    (double)W * H * spp / ms / 1e3);
        FILE *f = std::fopen(argv[2], "wb");               // PFM stores rows bottom-up, like the framebuffer
        if (!f) { std::perror(argv[2]); return 1; }
        std::fprintf(f, "PF\n%u %u\n-1.0\n", W, H);
        for (size_t p = 0; p < (size_t)W * H; p++) std::fwrite(&rgba[4 * p], sizeof(float), 3, f);
        std::fclose(f);
    } catch (const drt::Error &e) {
        std::fprintf(stderr, "drt error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
