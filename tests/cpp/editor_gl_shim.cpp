// editor_gl_shim.cpp -- the editor's display / "save png" statements (EditorLayer.cpp:85-96, 293-295, 317-318) over
// include/DustRayTracerGL.hpp, with a SOFTWARE texture standing in for OpenGL (this box has no GL context): the GL calls
// the shim makes are served by a table that stores the floats and converts on read-back the way the GL specification says
// (clamp to [0, 1], scale by 255, round to nearest).
//   editor_gl_shim out_basename [scene.glb]        without a scene: a synthetic gradient frame (no GPU needed)
#include <DustRayTracerGL.hpp>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>

namespace soft {
struct Tex { int w = 0, h = 0; std::vector<float> px; };
std::map<unsigned, Tex> textures;
unsigned bound = 0, next_name = 1;
int calls_sub = 0;
void GenTextures(int n, unsigned *t) { for (int i = 0; i < n; i++) { t[i] = next_name++; textures[t[i]] = Tex(); } }
void DeleteTextures(int n, const unsigned *t) { for (int i = 0; i < n; i++) textures.erase(t[i]); }
void BindTexture(unsigned, unsigned name) { bound = name; }
void TexParameteri(unsigned, unsigned, int) {}
void TexImage2D(unsigned, int, int internalformat, int w, int h, int, unsigned, unsigned, const void *) {
    if (internalformat != (int)drtgl::RGBA32F) std::abort();
    Tex &t = textures.at(bound); t.w = w; t.h = h; t.px.assign((size_t)w * h * 4, 0.f);
}
void TexSubImage2D(unsigned, int, int x, int y, int w, int h, unsigned format, unsigned type, const void *p) {
    Tex &t = textures.at(bound);
    if (x || y || w != t.w || h != t.h || format != drtgl::RGBA || type != drtgl::FLOAT) std::abort();
    std::memcpy(t.px.data(), p, t.px.size() * sizeof(float));
    calls_sub++;
}
void GetTexImage(unsigned, int, unsigned format, unsigned type, void *out) {
    const Tex &t = textures.at(bound);
    if (format != drtgl::RGBA || type != drtgl::UNSIGNED_BYTE) std::abort();
    unsigned char *o = static_cast<unsigned char *>(out);
    for (size_t i = 0; i < t.px.size(); i++) { float v = t.px[i]; v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v); o[i] = (unsigned char)std::lrintf(v * 255.0f); }
}
drtgl::GLApi api() {
    drtgl::GLApi a;
    a.GenTextures = GenTextures; a.DeleteTextures = DeleteTextures; a.BindTexture = BindTexture; a.TexParameteri = TexParameteri;
    a.TexImage2D = TexImage2D; a.TexSubImage2D = TexSubImage2D; a.GetTexImage = GetTexImage;
    return a;
}
}  // namespace soft

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s out_basename [scene.glb]\n", argv[0]); return 2; }
    try {
        drtgl::GLRenderTarget target(soft::api());
        if (argc >= 3) {                                   // OnAttach + OnUIRender, as tests/cpp/editor_calls.cpp, then display and save
            Scene scene;
            scene.loadGLTFmodel(argv[2]);
            BVHBuilder bvhbuilder;
            bvhbuilder.m_TargetLeafPrimitivesCount = 20; bvhbuilder.m_BinCount = 8;
            bvhbuilder.buildIterative(scene);
            Camera cam({ 3.6f, 1.25f, 0.f });
            cam.m_Forward_dir = { -1.f, 0.f, 0.f };
            Renderer m_Renderer;
            float m_LastRenderTime_ms = 0;
            for (int frame = 0; frame < 3; frame++) {      // three editor frames: Render, then the viewport texture is refreshed
                m_Renderer.ResizeBuffer(160, 90);
                m_Renderer.Render(&cam, scene, &m_LastRenderTime_ms);
                target.Update(m_Renderer);
            }
            std::printf("shown texture %u, %u x %u, sample %u, uploads %d\n", target.GetRenderTargetImage_name(), target.getBufferWidth(),
                        target.getBufferHeight(), m_Renderer.getSampleCount(), soft::calls_sub);
        } else {
            const uint32_t W = 37, H = 21;                 // synthetic frame: values below 0, inside and above 1
            std::vector<float> px((size_t)W * H * 4);
            for (uint32_t y = 0; y < H; y++)
                for (uint32_t x = 0; x < W; x++) {
                    float *p = &px[((size_t)y * W + x) * 4];
                    p[0] = (float)x / (float)(W - 1) * 1.5f - 0.25f; p[1] = (float)y / (float)(H - 1); p[2] = 0.5f; p[3] = 1.0f;
                }
            target.Update(px.data(), W, H);
            target.Update(px.data(), W, H);                // same size: no re-creation
            std::printf("shown texture %u, %u x %u, uploads %d\n", target.GetRenderTargetImage_name(), target.getBufferWidth(), target.getBufferHeight(), soft::calls_sub);
        }
        if (!target.saveImage(argv[1])) { std::perror(argv[1]); return 1; }        // "save png"
        std::printf("Image saved\n");
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
