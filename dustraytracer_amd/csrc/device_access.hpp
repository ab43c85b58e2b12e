// device_access.hpp -- device-side readers of the scene records (device_scene.hpp) and the small shading
// helpers shared by the kernels.  Citations are relative to /root/reference/DustRayTracer/src/.
#pragma once
#include "device_math.hpp"
#include "device_scene.hpp"

namespace drt {

DRT_DEV f2 interp_uv(const TriCold &c, f3 uvw) {                       // RayGen.cuh:116, AnyHit.cuh:20-22
    f2 r;
    r.x = uvw.x * c.uv[0][0] + uvw.y * c.uv[1][0] + uvw.z * c.uv[2][0];
    r.y = uvw.x * c.uv[0][1] + uvw.y * c.uv[1][1] + uvw.z * c.uv[2][1];
    return r;
}

DRT_DEV uint32_t texel_index(const TexDev &tex, f2 uv) {               // Texture.cu:35-36 / :65-66
    int x = (int)((uv.x - floorf(uv.x)) * tex.width);
    int y = (int)((uv.y - floorf(uv.y)) * tex.height);
    return (uint32_t)(y * tex.width + x);
}

// CLAMP (path_pool's hbm-scene and statistics builds): the texel index is held to the texture's padded extent -- width * height
// texels + the (width + 1) zero texels the packer appends for the latent out-of-bounds read of Texture.cu:35-49 -- which a valid
// header and any uv (finite, infinite or NaN) already guarantee: x <= width, y <= height.  Defence in depth, no effect on the value.
template <bool CLAMP>
DRT_DEV uint32_t texel_index_in(const TexDev &tex, f2 uv) {
    const uint32_t i = texel_index(tex, uv);
    return CLAMP ? min(i, (uint32_t)(tex.width * tex.height + tex.width)) : i;
}

template <bool CLAMP = false>
DRT_DEV f3 tex_get_pixel(const SceneView &sc, const TexDev &tex, f2 uv) {          // Texture.cu:33-58
    uint32_t i = texel_index_in<CLAMP>(tex, uv);
    float r = 0, g = 0, b = 255;
    if (tex.comps == 3 || tex.comps == 4) {
        const uint8_t *p = sc.texels + tex.offset + (size_t)i * (uint32_t)tex.comps;
        r = p[0]; g = p[1]; b = p[2];
    }
    f3 c = mk3(r / (float)255, g / (float)255, b / (float)255);
    return mk3(c.x * c.x, c.y * c.y, c.z * c.z);
}

template <bool CLAMP = false>
DRT_DEV float tex_get_alpha(const SceneView &sc, const TexDev &tex, f2 uv) {       // Texture.cu:60-75
    if (tex.comps < 4) return 1;
    uint32_t i = texel_index_in<CLAMP>(tex, uv);
    return sc.texels[tex.offset + (size_t)i * 4u + 3u] / (float)255;
}

DRT_DEV bool any_hit(const SceneView &sc, int prim, f3 uvw) {                      // AnyHit.cuh:8-28
    const TriCold cold = sc.tri_cold[prim];
    int tex_index = sc.mats[cold.material].tex;
    if (tex_index < 0) return true;
    const TexDev tex = sc.texs[tex_index];
    if (tex.comps < 4) return true;
    float alpha = tex_get_alpha(sc, tex, interp_uv(cold, uvw));
    return !(alpha < 1);
}

// One interior record = both child boxes + both child references.
struct ChildPair { f3 min1, max1, min2, max2; uint32_t ref1, ref2; };
DRT_DEV ChildPair load_children(const InnerNode *inner, uint32_t index) {
    const float4 *q = reinterpret_cast<const float4 *>(inner + index);
    float4 a = q[0], b = q[1], c = q[2];
    uint2 r = *reinterpret_cast<const uint2 *>(&q[3]);
    ChildPair p;
    p.min1 = mk3(a.x, a.y, a.z); p.max1 = mk3(a.w, b.x, b.y);
    p.min2 = mk3(b.z, b.w, c.x); p.max2 = mk3(c.y, c.z, c.w);
    p.ref1 = r.x; p.ref2 = r.y;
    return p;
}

struct TriTest { f3 v0, e1, e2; };
DRT_DEV TriTest load_tri(const TriHot *tris, int index) {
    const float4 *q = reinterpret_cast<const float4 *>(tris + index);
    float4 a = q[0], b = q[1];
    float c = reinterpret_cast<const float *>(&q[2])[0];
    TriTest t;
    t.v0 = mk3(a.x, a.y, a.z); t.e1 = mk3(a.w, b.x, b.y); t.e2 = mk3(b.z, b.w, c);
    return t;
}

// ---- Camera::GetRay, Scene/Camera.cu:98-122 (frame constants hoisted into FrameParams) ----
DRT_DEV Ray camera_get_ray(const FrameParams &fp, f2 uv, uint32_t &seed) {
    f2 offset;
    offset.x = random_float(seed) - 0.5f;
    offset.y = random_float(seed) - 0.5f;
    offset.x *= 0.0035f; offset.y *= 0.0035f;
    f3 pos = ld3(fp.cam_pos);
    f3 rorig = pos;
    if (fp.defocus) {
        f2 p = random_in_unit_disk(seed);
        rorig = pos + (p.x * ld3(fp.disk_u)) + (p.y * ld3(fp.disk_v));
    }
    f3 d = ld3(fp.fwd_focus) + ((uv.x + offset.x) * ld3(fp.horizontal)) + ((uv.y + offset.y) * ld3(fp.vertical)) - rorig + pos;
    return make_ray(rorig, normalize(d));
}

// ---- RenderKernel.cu:29-34 for one sample of a one-frame launch: accumulation_buffer[p] += c; texel = sum / frame index, alpha 1 ----
DRT_DEV void accumulate_and_resolve(const FrameParams &fp, uint32_t pixel, f3 c) {
    float *a = fp.accum + 3 * (size_t)pixel;
    const f3 acc = ld3(a) + c;
    a[0] = acc.x; a[1] = acc.y; a[2] = acc.z;
    const f3 out = acc / (float)fp.frame_first;
    reinterpret_cast<float4 *>(fp.rgba)[pixel] = make_float4(out.x, out.y, out.z, 1.0f);
}

}  // namespace drt
