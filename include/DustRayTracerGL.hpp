// DustRayTracerGL.hpp -- the editor's GL side of the render target, without CUDA-GL interop.
//
// The reference's Renderer owns an RGBA32F GL texture and the kernel writes it through a CUDA surface (Core/Renderer.cu:42-71
// texture creation, :84-94 interop mapping); the editor shows it with ImGui::Image(GetRenderTargetImage_name(), ..., {0,1}, {1,0})
// (Editor/EditorLayer.cpp:293-295) and "save png" reads it back with glGetTexImage(GL_RGBA, GL_UNSIGNED_BYTE) and writes it
// flipped (EditorLayer.cpp:23-31, 85-96; main.cpp stbi_flip_vertically_on_write).  ROCm has no such interop: GLRenderTarget
// keeps the texture on the editor's side and fills it from the renderer's host copy (glTexSubImage2D) after every Render.
//
// GL is reached through a table of function pointers (GLApi): GLApi::load() binds libGL at run time (no link-time
// dependency, so the header builds on a headless box); tests bind a software texture instead (tests/cpp/editor_gl_shim.cpp).
#pragma once
#include <dlfcn.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "DustRayTracer.hpp"

namespace drtgl {

typedef unsigned int GLuint_;
typedef unsigned char GLubyte_;
// the GL enumerants used (GL/gl.h, GL/glext.h values)
enum : unsigned { TEXTURE_2D = 0x0DE1, RGBA = 0x1908, RGBA32F = 0x8814, FLOAT = 0x1406, UNSIGNED_BYTE = 0x1401, TEXTURE_MIN_FILTER = 0x2801,
                  TEXTURE_MAG_FILTER = 0x2800, LINEAR = 0x2601 };

struct GLApi {
    void (*GenTextures)(int n, GLuint_ *textures) = nullptr;
    void (*DeleteTextures)(int n, const GLuint_ *textures) = nullptr;
    void (*BindTexture)(unsigned target, GLuint_ texture) = nullptr;
    void (*TexParameteri)(unsigned target, unsigned pname, int param) = nullptr;
    void (*TexImage2D)(unsigned target, int level, int internalformat, int width, int height, int border, unsigned format, unsigned type, const void *pixels) = nullptr;
    void (*TexSubImage2D)(unsigned target, int level, int xoffset, int yoffset, int width, int height, unsigned format, unsigned type, const void *pixels) = nullptr;
    void (*GetTexImage)(unsigned target, int level, unsigned format, unsigned type, void *pixels) = nullptr;
    bool complete() const { return GenTextures && DeleteTextures && BindTexture && TexParameteri && TexImage2D && TexSubImage2D && GetTexImage; }
    // the process's OpenGL (a current context is the caller's business, as in the editor)
    static GLApi load() {
        GLApi api;
        void *lib = nullptr;
        for (const char *name : { "libGL.so.1", "libOpenGL.so.0", "libGL.so" }) { lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
        if (!lib) throw std::runtime_error("no OpenGL library to bind (libGL.so.1)");
        api.GenTextures = reinterpret_cast<decltype(api.GenTextures)>(dlsym(lib, "glGenTextures"));
        api.DeleteTextures = reinterpret_cast<decltype(api.DeleteTextures)>(dlsym(lib, "glDeleteTextures"));
        api.BindTexture = reinterpret_cast<decltype(api.BindTexture)>(dlsym(lib, "glBindTexture"));
        api.TexParameteri = reinterpret_cast<decltype(api.TexParameteri)>(dlsym(lib, "glTexParameteri"));
        api.TexImage2D = reinterpret_cast<decltype(api.TexImage2D)>(dlsym(lib, "glTexImage2D"));
        api.TexSubImage2D = reinterpret_cast<decltype(api.TexSubImage2D)>(dlsym(lib, "glTexSubImage2D"));
        api.GetTexImage = reinterpret_cast<decltype(api.GetTexImage)>(dlsym(lib, "glGetTexImage"));
        if (!api.complete()) throw std::runtime_error("the OpenGL library lacks an entry point");
        return api;
    }
};

// Minimal PNG writer (RGBA8, stored deflate blocks): enough for a viewer to open the result.
inline uint32_t crc32_of(const uint8_t *p, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    if (!table[1]) for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
inline void put_chunk(FILE *f, const char *tag, const std::vector<uint8_t> &body) {
    uint8_t len[4] = { (uint8_t)(body.size() >> 24), (uint8_t)(body.size() >> 16), (uint8_t)(body.size() >> 8), (uint8_t)body.size() };
    std::fwrite(len, 1, 4, f);
    std::vector<uint8_t> buf(tag, tag + 4);
    buf.insert(buf.end(), body.begin(), body.end());
    std::fwrite(buf.data(), 1, buf.size(), f);
    uint32_t c = crc32_of(buf.data(), buf.size());
    uint8_t cb[4] = { (uint8_t)(c >> 24), (uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c };
    std::fwrite(cb, 1, 4, f);
}
inline bool write_png_rgba8(const char *path, uint32_t w, uint32_t h, const std::vector<uint8_t> &rgba_top_down) {
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


// The texture the viewport shows (Renderer::m_RenderTargetTexture_name, Renderer.hpp:32) and what the editor does with it.
class GLRenderTarget {
public:
    explicit GLRenderTarget(const GLApi &api) : gl(api) { if (!gl.complete()) throw std::runtime_error("incomplete GLApi"); }
    ~GLRenderTarget() { if (m_name) gl.DeleteTextures(1, &m_name); }
    GLRenderTarget(const GLRenderTarget &) = delete;
    GLRenderTarget &operator=(const GLRenderTarget &) = delete;

    GLuint_ &GetRenderTargetImage_name() { return m_name; }               // Renderer.hpp:21: what ImGui::Image is given (EditorLayer.cpp:293-295)
    uint32_t getBufferWidth() const { return m_width; }
    uint32_t getBufferHeight() const { return m_height; }

    // after Renderer::Render: the host copy of the frame goes into the texture (Renderer.cu:42-71 creates it RGBA32F, linear filtered)
    void Update(const float *rgba32f, uint32_t width, uint32_t height) {
        if (width != m_width || height != m_height || !m_name) {
            if (m_name) gl.DeleteTextures(1, &m_name);
            gl.GenTextures(1, &m_name);
            gl.BindTexture(TEXTURE_2D, m_name);
            gl.TexParameteri(TEXTURE_2D, TEXTURE_MIN_FILTER, LINEAR);
            gl.TexParameteri(TEXTURE_2D, TEXTURE_MAG_FILTER, LINEAR);
            gl.TexImage2D(TEXTURE_2D, 0, (int)RGBA32F, (int)width, (int)height, 0, RGBA, FLOAT, nullptr);
            m_width = width; m_height = height;
        }
        gl.BindTexture(TEXTURE_2D, m_name);
        gl.TexSubImage2D(TEXTURE_2D, 0, 0, 0, (int)width, (int)height, RGBA, FLOAT, rgba32f);
        gl.BindTexture(TEXTURE_2D, 0);
    }
    void Update(Renderer &renderer) {
        m_pixels.resize((size_t)renderer.getBufferWidth() * renderer.getBufferHeight() * 4);
        renderer.ReadRenderTarget(m_pixels.data());
        Update(m_pixels.data(), renderer.getBufferWidth(), renderer.getBufferHeight());
    }
    // "save png" (EditorLayer.cpp:85-96): read the texture back as RGBA8 (GL clamps to [0, 1] and scales to 8 bits) ...
    std::vector<GLubyte_> ReadBackRGBA8() {
        std::vector<GLubyte_> frame_data((size_t)m_width * m_height * 4);
        gl.BindTexture(TEXTURE_2D, m_name);
        gl.GetTexImage(TEXTURE_2D, 0, RGBA, UNSIGNED_BYTE, frame_data.data());
        gl.BindTexture(TEXTURE_2D, 0);
        return frame_data;
    }
    // ... and write "<filename>.png" with the rows flipped (EditorLayer.cpp:23-31; stbi_flip_vertically_on_write(true), Application.cpp)
    bool saveImage(const char *filename) {
        const std::vector<GLubyte_> data = ReadBackRGBA8();
        std::vector<uint8_t> flipped(data.size());
        for (uint32_t y = 0; y < m_height; y++)
            std::copy(data.begin() + (size_t)(m_height - 1 - y) * m_width * 4, data.begin() + (size_t)(m_height - y) * m_width * 4, flipped.begin() + (size_t)y * m_width * 4);
        return write_png_rgba8((std::string(filename) + ".png").c_str(), m_width, m_height, flipped);
    }

private:
    GLApi gl;
    GLuint_ m_name = 0;
    uint32_t m_width = 0, m_height = 0;
    std::vector<float> m_pixels;
};

}  // namespace drtgl
