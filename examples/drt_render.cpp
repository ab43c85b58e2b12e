// drt_render.cpp -- headless use of the reference-shaped C++ API: scene.glb -> RGBA32F -> PFM or PNG file.
// (.png output reproduces the editor's "save png": 8-bit clamp of the GL read-back + vertical flip, EditorLayer.cpp:23-31,85-96)
//   g++ -std=c++17 -Iinclude examples/drt_render.cpp -Ldustraytracer_amd -ldrt_hip -Wl,-rpath,$PWD/dustraytracer_amd -o drt_render
//   ./drt_render models/cornell_box.glb out.pfm 1920 1080 8 8  3.6 1.25 0  -1 0 0
//   DRT_DEVICES=0,1,2,3,4,5,6,7 ./drt_render models/room.glb out.pfm 3840 2160 64 16  0 1.4 2  0 0 -1     (all GPUs of the node: stripes + RCCL gather)
#include <DustRayTracer.hpp>
#include <DustRayTracerGL.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using drtgl::write_png_rgba8;      // the minimal PNG writer lives in DustRayTracerGL.hpp (the editor shim's "save png")

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
        std::vector<int> devices;                         // DRT_DEVICES=0,1,...: several GPUs of the node behind the same Renderer calls
        if (const char *list = std::getenv("DRT_DEVICES"))
            for (const char *p = list; *p;) { devices.push_back(std::atoi(p)); while (*p && *p != ',') p++; if (*p == ',') p++; }
        if (devices.empty()) devices.push_back(0);
        Renderer renderer(devices);
        renderer.m_RendererSettings.ray_bounce_limit = std::atoi(argv[6]);
        renderer.m_RendererSettings.max_samples = (int)spp + 1;
        renderer.ResizeBuffer(W, H);
        float ms = 0;
        renderer.RenderBatch(&cam, scene, spp, &ms);
        std::vector<float> rgba((size_t)W * H * 4);
        renderer.ReadRenderTarget(rgba.data());
        std::printf("%zu triangles, %u x %u, %u spp on %d GPU%s: %.3f ms (%.1f Msamples/s)\n", scene.trianglesCount(), W, H, spp,
                    renderer.deviceCount(), renderer.deviceCount() > 1 ? "s" : "", ms, (double)W * H * spp / ms / 1e3);
        const std::string out(argv[2]);
        if (out.size() > 4 && out.compare(out.size() - 4, 4, ".png") == 0) {
            // glGetTexImage(GL_RGBA, GL_UNSIGNED_BYTE) clamps to [0,1] and rounds to 8 bits; stbi_flip_vertically_on_write(true)
            std::vector<uint8_t> bytes((size_t)W * H * 4);
            for (uint32_t y = 0; y < H; y++)
                for (uint32_t x = 0; x < W; x++)
                    for (int c = 0; c < 4; c++) {
                        float v = rgba[((size_t)(H - 1 - y) * W + x) * 4 + c];
                        v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
                        bytes[((size_t)y * W + x) * 4 + c] = (uint8_t)(v * 255.0f + 0.5f);
                    }
            if (!write_png_rgba8(argv[2], W, H, bytes)) { std::perror(argv[2]); return 1; }
        } else {
            FILE *f = std::fopen(argv[2], "wb");           // PFM stores rows bottom-up, like the framebuffer
            if (!f) { std::perror(argv[2]); return 1; }
            std::fprintf(f, "PF\n%u %u\n-1.0\n", W, H);
            for (size_t p = 0; p < (size_t)W * H; p++) std::fwrite(&rgba[4 * p], sizeof(float), 3, f);
            std::fclose(f);
        }
    } catch (const drt::Error &e) {
        std::fprintf(stderr, "drt error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
