// drt_render.cpp -- headless use of the reference-shaped C++ API: scene.glb -> RGBA32F -> PFM or PNG file.
// (.png output reproduces the editor's "save png": 8-bit clamp of the GL read-back + vertical flip, EditorLayer.cpp:23-31,85-96)
//   g++ -std=c++17 -Iinclude examples/drt_render.cpp -Ldustraytracer_amd -ldrt_hip -Wl,-rpath,$PWD/dustraytracer_amd -o drt_render
//   ./drt_render models/cornell_box.glb out.pfm 1920 1080 8 8  3.6 1.25 0  -1 0 0
//   DRT_DEVICES=0,1,2,3,4,5,6,7 ./drt_render models/room.glb out.pfm 3840 2160 64 16  0 1.4 2  0 0 -1     (all GPUs of the node: stripes + RCCL gather)
#include <DustRayTracer.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// Minimal PNG writer (RGBA8, stored deflate blocks): enough for a viewer to open the result.
static uint32_t crc32_of(const uint8_t *p, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    if (!table[1]) for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
static void put_chunk(FILE *f, const char *tag, const std::vector<uint8_t> &body) {
    uint8_t len[4] = { (uint8_t)(body.size() >> 24), (uint8_t)(body.size() >> 16), (uint8_t)(body.size() >> 8), (uint8_t)body.size() };
    std::fwrite(len, 1, 4, f);
    std::vector<uint8_t> buf(tag, tag + 4);
    buf.insert(buf.end(), body.begin(), body.end());
    std::fwrite(buf.data(), 1, buf.size(), f);
    uint32_t c = crc32_of(buf.data(), buf.size());
    uint8_t cb[4] = { (uint8_t)(c >> 24), (uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c };
    std::fwrite(cb, 1, 4, f);
}
static bool write_png_rgba8(const char *path, uint32_t w, uint32_t h, const std::vector<uint8_t> &rgba_top_down) {
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    std::fwrite(sig, 1, 8, f);
    std::vector<uint8_t> ihdr = { (uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w,
                                  (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h, 8, 6, 0, 0, 0 };
    put_chunk(f, "IHDR", ihdr);
    std::vector<uint8_t> raw;                       // filter byte 0 + scanline
    raw.reserve((size_t)h * (w * 4 + 1));
    for (uint32_t y = 0; y < h; y++) { raw.push_back(0); raw.insert(raw.end(), rgba_top_down.begin() + (size_t)y * w * 4, rgba_top_down.begin() + (size_t)(y + 1) * w * 4); }
    std::vector<uint8_t> z = { 0x78, 0x01 };        // zlib header, then stored blocks of <= 65535 bytes
    uint32_t a = 1, b = 0;
    for (uint8_t v : raw) { a = (a + v) % 65521u; b = (b + a) % 65521u; }
    for (size_t off = 0; off < raw.size();) {
        size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)n); z.push_back((uint8_t)(n >> 8)); z.push_back((uint8_t)~n); z.push_back((uint8_t)(~n >> 8));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        off += n;
    }
    uint32_t adler = (b << 16) | a;
    z.push_back((uint8_t)(adler >> 24)); z.push_back((uint8_t)(adler >> 16)); z.push_back((uint8_t)(adler >> 8)); z.push_back((uint8_t)adler);
    put_chunk(f, "IDAT", z);
    put_chunk(f, "IEND", {});
    return std::fclose(f) == 0;
}

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
