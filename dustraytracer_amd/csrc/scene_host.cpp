// scene_host.cpp -- glTF flattening + binned-SAH BVH build + device packing (host prep for the hot path).
//
// Written from scratch; reproduces the *decisions* of the reference so that the triangle order, the node
// array and therefore the rendered image are identical:
//   Scene::loadGLTFmodel / parseMesh / loadMaterials / loadTextures   Core/Scene/Scene.cu:59-317
//   BVHBuilder::buildIterative / makePartition / binToShallowNodes / binToNodes   Core/BVH/BVHBuilder.cu:11-346
// Compile with -ffp-contract=off: every product and sum below rounds on its own, like the reference's
// host code built without FMA.
#include "scene_host.hpp"
#include "bvh_build_device.hpp"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

#include "json_min.hpp"
#include "png_decode.hpp"

namespace drt {

uint64_t HostScene::next_revision() {
    static std::atomic<uint64_t> counter{ 0 };
    return ++counter;
}

V3 normalize(V3 v) {
    float inv_len = 1.0f / sqrtf(dot(v, v));
    return v * inv_len;
}

static inline V3 ld(const float *p) { return V3{ p[0], p[1], p[2] }; }
static inline void st(float *p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

void HostScene::clear() {
    triangles.clear(); materials.clear(); textures.clear(); meshes.clear(); nodes.clear();
    revision = next_revision();
}

// ---------------------------------------------------------------------------------------------
// Triangle assembly (Scene.cu:272-302, Triangle.cuh:9-12)
// ---------------------------------------------------------------------------------------------
static drt_triangle make_triangle(const float *pos, const float *nrm, const float *uv, int32_t material) {
    drt_triangle t;
    std::memset(&t, 0, sizeof t);
    V3 P[3], N[3];
    for (int k = 0; k < 3; k++) {
        P[k] = ld(pos + 3 * k);
        N[k] = ld(nrm + 3 * k);
        st(t.vertex[k].position, P[k]);
        st(t.vertex[k].normal, N[k]);
        t.vertex[k].uv[0] = uv[2 * k];
        t.vertex[k].uv[1] = uv[2 * k + 1];
    }
    V3 face = cross(P[1] - P[0], P[2] - P[0]);
    V3 avg = (N[0] + N[1] + N[2]) / 3;
    float ndot = dot(face, avg);
    V3 oriented = (ndot < 0.0f) ? v3(-face.x, -face.y, -face.z) : face;
    st(t.face_normal, normalize(oriented));
    st(t.centroid, (P[0] + P[1] + P[2]) / 3);
    t.material = material;
    return t;
}

void HostScene::set_geometry(const float *pos, const float *nrm, const float *uv, const int32_t *mat, int32_t n_tris) {
    drt_mesh mesh;
    mesh.primitives_offset = (int32_t)triangles.size();
    for (int32_t i = 0; i < n_tris; i++)
        triangles.push_back(make_triangle(pos + 9 * (size_t)i, nrm + 9 * (size_t)i, uv + 6 * (size_t)i, mat[i]));
    mesh.tris_count = n_tris;
    meshes.push_back(mesh);
    nodes.clear();
    revision = next_revision();
}

// ---------------------------------------------------------------------------------------------
// GLB container + the JSON fields the reference consumes
// ---------------------------------------------------------------------------------------------
static std::vector<uint8_t> read_file(const char *path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw IoError(std::string("cannot open ") + path);
    f.seekg(0, std::ios::end);
    std::streamoff n = f.tellg();
    f.seekg(0, std::ios::beg);
    std::vector<uint8_t> data((size_t)std::max<std::streamoff>(n, 0));
    if (n > 0 && !f.read((char *)data.data(), n)) throw IoError(std::string("cannot read ") + path);
    return data;
}

// data:<mime>;base64,<payload>  (RFC 2397, base64 only -- what glTF exporters write)
static std::vector<uint8_t> decode_data_uri(const std::string &uri) {
    const size_t comma = uri.find(',');
    if (comma == std::string::npos || uri.find(";base64") == std::string::npos || uri.find(";base64") > comma)
        throw UnsupportedError("data: URI that is not base64");
    std::vector<uint8_t> out;
    out.reserve((uri.size() - comma) * 3 / 4);
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = comma + 1; i < uri.size(); i++) {
        const char c = uri[i];
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A';
        else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
        else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62;
        else if (c == '/' || c == '_') v = 63;
        else if (c == '=') break;
        else continue;                                 // whitespace
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); }
    }
    return out;
}

static uint32_t le32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

struct BufferView { int64_t buffer = 0, offset = 0, length = 0, stride = 0; };

// ---------------------------------------------------------------------------------------------
// Opt-in: geometry as the glTF 2.0 specification defines it (drt_scene_load_gltf_ex, DRT_LOAD_STRICT).  What the
// reference's loader assumes away (Scene.cu:120-200) is honoured here: the scene graph and node transforms, accessor
// byteOffset / componentType / bufferView.byteStride, u8 / u16 / u32 or absent indices, nodes without a mesh, a mesh used
// by several nodes, missing NORMAL / TEXCOORD_0 / material.  Triangle assembly (centroid, face normal turned towards the
// averaged vertex normal) is the reference's, so a file that satisfies the reference's assumptions loads identically.
// ---------------------------------------------------------------------------------------------
namespace {

struct Accessor {
    const uint8_t *base = nullptr;
    size_t stride = 0, count = 0;
    int component = 0, width = 0;
    bool normalized = false;
    double get(size_t i, int k) const {
        const uint8_t *p = base + i * stride;
        switch (component) {
        case 5126: { float v; std::memcpy(&v, p + 4 * k, 4); return v; }
        case 5121: { uint8_t v = p[k]; return normalized ? v / 255.0 : v; }
        case 5123: { uint16_t v; std::memcpy(&v, p + 2 * k, 2); return normalized ? v / 65535.0 : v; }
        case 5125: { uint32_t v; std::memcpy(&v, p + 4 * k, 4); return v; }
        case 5120: { int8_t v; std::memcpy(&v, p + k, 1); return normalized ? std::max(v / 127.0, -1.0) : v; }
        case 5122: { int16_t v; std::memcpy(&v, p + 2 * k, 2); return normalized ? std::max(v / 32767.0, -1.0) : v; }
        default: return 0;
        }
    }
};

Accessor open_accessor(const JsonValue &root, const std::vector<std::pair<const uint8_t *, size_t>> &buffers,
                       const std::vector<BufferView> &views, int64_t index, int want_width) {
    const JsonValue &jacc = root.at("accessors");
    if (index < 0 || (size_t)index >= jacc.size()) throw std::runtime_error("accessor index out of range");
    const JsonValue &a = jacc.at((size_t)index);
    if (a.has("sparse")) throw UnsupportedError("sparse accessors");
    const int64_t bv = a.at("bufferView").as_int(-1);
    if (bv < 0 || (size_t)bv >= views.size()) throw std::runtime_error("accessor without bufferView");
    const std::string type = a.at("type").as_string();
    Accessor acc;
    acc.width = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : 0;
    if (acc.width != want_width) throw std::runtime_error("accessor type does not fit its use");
    acc.component = (int)a.at("componentType").as_int(0);
    const size_t csize = acc.component == 5126 || acc.component == 5125 ? 4 : acc.component == 5123 || acc.component == 5122 ? 2
                       : acc.component == 5121 || acc.component == 5120 ? 1 : 0;
    if (!csize) throw std::runtime_error("unknown accessor componentType");
    acc.normalized = a.at("normalized").kind == JsonValue::Bool && a.at("normalized").b;
    // Every number comes from the file: negative values must not turn into huge size_t's, and the range check must not wrap.
    const int64_t count = a.at("count").as_int(0), byte_offset = a.at("byteOffset").as_int(0);
    const BufferView &v = views[(size_t)bv];
    if (count < 0 || byte_offset < 0 || v.stride < 0) throw std::runtime_error("accessor with a negative count, byteOffset or byteStride");
    acc.count = (size_t)count;
    const size_t elem = csize * (size_t)acc.width;
    acc.stride = v.stride > 0 ? (size_t)v.stride : elem;
    const size_t offset = (size_t)byte_offset, length = (size_t)v.length;
    // last element ends at offset + stride*(count-1) + elem <= length, written without a product that can overflow
    if (acc.count && (offset > length || elem > length - offset || (acc.count - 1) > (length - offset - elem) / acc.stride))
        throw std::runtime_error("accessor runs past its bufferView");
    acc.base = buffers[(size_t)v.buffer].first + v.offset + offset;
    return acc;
}

struct Mat4 { double m[16]; };      // column-major, as glTF stores it
Mat4 identity() { Mat4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1; return r; }
Mat4 mul(const Mat4 &a, const Mat4 &b) {
    Mat4 r{};
    for (int c = 0; c < 4; c++)
        for (int row = 0; row < 4; row++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += a.m[k * 4 + row] * b.m[c * 4 + k];
            r.m[c * 4 + row] = s;
        }
    return r;
}
Mat4 local_matrix(const JsonValue &node) {
    if (node.has("matrix") && node.at("matrix").size() == 16) {
        Mat4 r;
        for (int i = 0; i < 16; i++) r.m[i] = node.at("matrix").at((size_t)i).as_double(0);
        return r;
    }
    double t[3] = { 0, 0, 0 }, q[4] = { 0, 0, 0, 1 }, sc[3] = { 1, 1, 1 };
    if (node.at("translation").size() == 3) for (int i = 0; i < 3; i++) t[i] = node.at("translation").at((size_t)i).as_double(0);
    if (node.at("rotation").size() == 4) for (int i = 0; i < 4; i++) q[i] = node.at("rotation").at((size_t)i).as_double(i == 3);
    if (node.at("scale").size() == 3) for (int i = 0; i < 3; i++) sc[i] = node.at("scale").at((size_t)i).as_double(1);
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    Mat4 r = identity();
    r.m[0] = (1 - 2 * (y * y + z * z)) * sc[0]; r.m[1] = (2 * (x * y + z * w)) * sc[0]; r.m[2] = (2 * (x * z - y * w)) * sc[0];
    r.m[4] = (2 * (x * y - z * w)) * sc[1]; r.m[5] = (1 - 2 * (x * x + z * z)) * sc[1]; r.m[6] = (2 * (y * z + x * w)) * sc[1];
    r.m[8] = (2 * (x * z + y * w)) * sc[2]; r.m[9] = (2 * (y * z - x * w)) * sc[2]; r.m[10] = (1 - 2 * (x * x + y * y)) * sc[2];
    r.m[12] = t[0]; r.m[13] = t[1]; r.m[14] = t[2];
    return r;
}
// inverse transpose of the upper 3x3 (cofactor matrix / det): how normals transform
void normal_matrix(const Mat4 &w, double n[9]) {
    const double a = w.m[0], b = w.m[4], c = w.m[8], d = w.m[1], e = w.m[5], f = w.m[9], g = w.m[2], h = w.m[6], i = w.m[10];
    const double co[9] = { e * i - f * h, f * g - d * i, d * h - e * g, c * h - b * i, a * i - c * g, b * g - a * h, b * f - c * e, c * d - a * f, a * e - b * d };
    const double det = a * co[0] + b * co[1] + c * co[2];
    const double s = det != 0 ? 1.0 / det : 1.0;
    for (int k = 0; k < 9; k++) n[k] = co[k] * s;       // n[3*row + col] = cofactor(row, col) / det  == (M^-1)^T
}

}  // namespace

static void load_strict_geometry(const JsonValue &root, const std::vector<std::pair<const uint8_t *, size_t>> &buffers,
                                 const std::vector<BufferView> &views, HostScene &out) {
    const JsonValue &jnodes = root.at("nodes");
    const JsonValue &jmeshes = root.at("meshes");
    int32_t default_material = -1;
    // roots: the default scene's (or scene 0's) nodes; without scenes, every node that is nobody's child
    std::vector<size_t> roots;
    const JsonValue &jscenes = root.at("scenes");
    if (jscenes.size() > 0) {
        size_t which = (size_t)std::max<long long>(0, root.at("scene").as_int(0));
        if (which >= jscenes.size()) which = 0;
        const JsonValue &list = jscenes.at(which).at("nodes");
        for (size_t i = 0; i < list.size(); i++) roots.push_back((size_t)list.at(i).as_int(0));
    } else {
        std::vector<char> is_child(jnodes.size(), 0);
        for (size_t n = 0; n < jnodes.size(); n++) {
            const JsonValue &ch = jnodes.at(n).at("children");
            for (size_t i = 0; i < ch.size(); i++) { const long long c = ch.at(i).as_int(-1); if (c >= 0 && (size_t)c < jnodes.size()) is_child[(size_t)c] = 1; }
        }
        for (size_t n = 0; n < jnodes.size(); n++) if (!is_child[n]) roots.push_back(n);
    }
    struct Item { size_t node; Mat4 world; int depth; };
    std::vector<Item> stack;
    for (size_t i = roots.size(); i-- > 0;) stack.push_back(Item{ roots[i], identity(), 0 });
    while (!stack.empty()) {
        const Item it = stack.back();
        stack.pop_back();
        if (it.node >= jnodes.size()) throw std::runtime_error("node index out of range");
        if (it.depth > 256) throw std::runtime_error("node hierarchy deeper than 256 levels (a cycle?)");
        const JsonValue &node = jnodes.at(it.node);
        const Mat4 world = mul(it.world, local_matrix(node));
        const JsonValue &ch = node.at("children");
        for (size_t i = ch.size(); i-- > 0;) stack.push_back(Item{ (size_t)ch.at(i).as_int(0), world, it.depth + 1 });
        const long long mesh_index = node.at("mesh").as_int(-1);
        if (mesh_index < 0) continue;                               // cameras, lights, empties
        if ((size_t)mesh_index >= jmeshes.size()) throw std::runtime_error("mesh index out of range");
        double nm[9];
        normal_matrix(world, nm);
        bool untransformed = true;                                  // identity: hand the file's floats through untouched
        for (int k = 0; k < 16; k++) untransformed = untransformed && world.m[k] == (k % 5 == 0 ? 1.0 : 0.0);
        std::vector<float> pos, nrm, uv;
        std::vector<int32_t> mat;
        const JsonValue &prims = jmeshes.at((size_t)mesh_index).at("primitives");
        for (size_t p = 0; p < prims.size(); p++) {
            const JsonValue &prim = prims.at(p);
            const long long mode = prim.at("mode").as_int(4);
            if (mode < 4) continue;                                 // points and lines are not renderable here
            if (mode != 4) throw UnsupportedError("triangle strips / fans");
            const JsonValue &attrs = prim.at("attributes");
            if (!attrs.has("POSITION")) continue;
            const Accessor ap = open_accessor(root, buffers, views, attrs.at("POSITION").as_int(-1), 3);
            const bool has_n = attrs.has("NORMAL"), has_t = attrs.has("TEXCOORD_0");
            Accessor an, at_;
            if (has_n) an = open_accessor(root, buffers, views, attrs.at("NORMAL").as_int(-1), 3);
            if (has_t) at_ = open_accessor(root, buffers, views, attrs.at("TEXCOORD_0").as_int(-1), 2);
            Accessor ai;
            const bool indexed = prim.has("indices");
            if (indexed) ai = open_accessor(root, buffers, views, prim.at("indices").as_int(-1), 1);
            const size_t n_idx = indexed ? ai.count : ap.count;
            int32_t material = (int32_t)prim.at("material").as_int(-1);
            if (material < 0) {                                     // the specification's default material: white, untextured
                if (default_material < 0) {
                    drt_material m;
                    std::memset(&m, 0, sizeof m);
                    m.albedo[0] = m.albedo[1] = m.albedo[2] = 1; m.albedo_tex = -1; m.roughness = 1; m.metallic = 1; m.refractive_index = 1.45f;
                    default_material = (int32_t)out.materials.size();
                    out.materials.push_back(m);
                }
                material = default_material;
            }
            for (size_t t = 0; t + 3 <= n_idx; t += 3) {
                double wp[3][3];
                for (int k = 0; k < 3; k++) {
                    const size_t v = indexed ? (size_t)ai.get(t + (size_t)k, 0) : t + (size_t)k;
                    if (v >= ap.count || (has_n && v >= an.count) || (has_t && v >= at_.count)) throw std::runtime_error("vertex index out of range");
                    const double x = ap.get(v, 0), y = ap.get(v, 1), z = ap.get(v, 2);
                    if (untransformed) { wp[k][0] = x; wp[k][1] = y; wp[k][2] = z; }
                    else for (int r = 0; r < 3; r++) wp[k][r] = world.m[r] * x + world.m[4 + r] * y + world.m[8 + r] * z + world.m[12 + r];
                    for (int r = 0; r < 3; r++) pos.push_back((float)wp[k][r]);
                    if (has_t) { uv.push_back((float)at_.get(v, 0)); uv.push_back((float)at_.get(v, 1)); }
                    else { uv.push_back(0.f); uv.push_back(0.f); }
                    if (has_n) {
                        const double nx = an.get(v, 0), ny = an.get(v, 1), nz = an.get(v, 2);
                        if (untransformed) { nrm.push_back((float)nx); nrm.push_back((float)ny); nrm.push_back((float)nz); continue; }
                        double w3[3];
                        for (int r = 0; r < 3; r++) w3[r] = nm[3 * r] * nx + nm[3 * r + 1] * ny + nm[3 * r + 2] * nz;
                        const double len = std::sqrt(w3[0] * w3[0] + w3[1] * w3[1] + w3[2] * w3[2]);
                        for (int r = 0; r < 3; r++) nrm.push_back((float)(len > 0 ? w3[r] / len : w3[r]));
                    }
                }
                if (!has_n) {                                       // flat shading: the geometric normal (counter-clockwise front)
                    const double e1[3] = { wp[1][0] - wp[0][0], wp[1][1] - wp[0][1], wp[1][2] - wp[0][2] };
                    const double e2[3] = { wp[2][0] - wp[0][0], wp[2][1] - wp[0][1], wp[2][2] - wp[0][2] };
                    double g[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
                    const double len = std::sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
                    for (int k = 0; k < 3; k++) for (int r = 0; r < 3; r++) nrm.push_back((float)(len > 0 ? g[r] / len : 0.0));
                }
                mat.push_back(material);
            }
        }
        for (int32_t m : mat)
            if (m < 0 || (size_t)m >= out.materials.size()) throw std::runtime_error("primitive.material out of range");
        if (!mat.empty()) out.set_geometry(pos.data(), nrm.data(), uv.data(), mat.data(), (int32_t)mat.size());
    }
}

void HostScene::load_gltf(const char *path, bool strict) {
    std::string spath(path ? path : "");
    size_t dot_at = spath.find_last_of('.');
    std::string ext = dot_at == std::string::npos ? "" : spath.substr(dot_at + 1);
    const bool is_binary = ext == "glb";             // Scene.cu:32-41: ".glb" = binary container, anything else = ASCII glTF
    const size_t slash_at = spath.find_last_of("/\\");
    const std::string base_dir = slash_at == std::string::npos ? std::string() : spath.substr(0, slash_at + 1);

    std::vector<uint8_t> blob = read_file(path);
    const uint8_t *json_p = nullptr, *bin_p = nullptr;
    size_t json_n = 0, bin_n = 0;
    if (is_binary) {
        if (blob.size() < 20 || std::memcmp(blob.data(), "glTF", 4) != 0) throw std::runtime_error("not a GLB file");
        if (le32(blob.data() + 4) != 2) throw UnsupportedError("glTF container version != 2");
        size_t total = std::min<size_t>(le32(blob.data() + 8), blob.size());
        for (size_t off = 12; off + 8 <= total;) {
            uint32_t clen = le32(blob.data() + off), ctype = le32(blob.data() + off + 4);
            if ((size_t)clen > total - off - 8) throw std::runtime_error("GLB chunk runs past end of file");
            if (ctype == 0x4E4F534Au && !json_p) { json_p = blob.data() + off + 8; json_n = clen; }
            else if (ctype == 0x004E4942u && !bin_p) { bin_p = blob.data() + off + 8; bin_n = clen; }
            off += 8 + (size_t)clen + ((4 - clen % 4) % 4);
        }
        if (!json_p) throw std::runtime_error("GLB has no JSON chunk");
    } else {
        json_p = blob.data(); json_n = blob.size();
    }
    JsonValue root = JsonParser((const char *)json_p, json_n).parse();

    // buffers: the GLB-embedded buffer is the BIN chunk, truncated to byteLength; a buffer with a uri is a file next to the
    // .gltf (what tinygltf's LoadASCIIFromFile resolves it against) or a base64 data: URI
    std::vector<std::pair<const uint8_t *, size_t>> buffers;
    std::vector<std::vector<uint8_t>> owned;          // external buffers live here until the end of the load
    const JsonValue &jbuffers = root.at("buffers");
    owned.reserve(jbuffers.size());
    for (size_t i = 0; i < jbuffers.size(); i++) {
        const JsonValue &b = jbuffers.at(i);
        size_t len = (size_t)b.at("byteLength").as_int(0);
        if (b.has("uri")) {
            const std::string uri = b.at("uri").as_string();
            owned.push_back(uri.compare(0, 5, "data:") == 0 ? decode_data_uri(uri) : read_file((base_dir + uri).c_str()));
            if (len > owned.back().size()) throw std::runtime_error("buffer.byteLength exceeds the external buffer");
            buffers.emplace_back(owned.back().data(), len);
            continue;
        }
        if (!is_binary) throw std::runtime_error("ASCII glTF buffer without a uri");
        if (len > bin_n) throw std::runtime_error("buffer.byteLength exceeds the BIN chunk");
        buffers.emplace_back(bin_p, len);
    }
    std::vector<BufferView> views;
    const JsonValue &jviews = root.at("bufferViews");
    for (size_t i = 0; i < jviews.size(); i++) {
        BufferView v;
        v.buffer = jviews.at(i).at("buffer").as_int(0);
        v.offset = jviews.at(i).at("byteOffset").as_int(0);
        v.length = jviews.at(i).at("byteLength").as_int(0);
        v.stride = jviews.at(i).at("byteStride").as_int(0);
        if (v.buffer < 0 || (size_t)v.buffer >= buffers.size() || v.offset < 0 || v.length < 0 || v.stride < 0 ||
            (uint64_t)v.offset > buffers[(size_t)v.buffer].second || (uint64_t)v.length > buffers[(size_t)v.buffer].second - (uint64_t)v.offset)
            throw std::runtime_error("bufferView out of range");
        views.push_back(v);
    }
    const JsonValue &jaccessors = root.at("accessors");
    auto accessor_view = [&](int64_t accessor) -> const BufferView & {
        if (accessor < 0 || (size_t)accessor >= jaccessors.size()) throw std::runtime_error("accessor index out of range");
        int64_t bv = jaccessors.at((size_t)accessor).at("bufferView").as_int(-1);
        if (bv < 0 || (size_t)bv >= views.size()) throw std::runtime_error("accessor without bufferView");
        return views[(size_t)bv];
    };

    HostScene fresh;

    // loadTextures (Scene.cu:88-117): one texture per IMAGE, decoded from its bufferView
    const JsonValue &jimages = root.at("images");
    for (size_t i = 0; i < jimages.size(); i++) {
        std::vector<uint8_t> file;                    // ASCII branch: the image is a file
        const uint8_t *p = nullptr;
        size_t n = 0;
        if (is_binary) {                              // Scene.cu:100-104
            int64_t bv = jimages.at(i).at("bufferView").as_int(-1);
            if (bv < 0 || (size_t)bv >= views.size()) throw UnsupportedError("GLB image without bufferView (the reference indexes bufferViews[-1] here)");
            const BufferView &v = views[(size_t)bv];
            p = buffers[(size_t)v.buffer].first + v.offset; n = (size_t)v.length;
        } else {
            // Scene.cu:90,111: the reference opens "../models/" + uri relative to the process's working directory.  That
            // path is tried first; a file next to the .gltf (where the uri actually points) is the fallback.
            const std::string uri = jimages.at(i).at("uri").as_string();
            if (uri.empty()) throw UnsupportedError("ASCII glTF image without a uri");
            if (uri.compare(0, 5, "data:") == 0) file = decode_data_uri(uri);
            else {
                try { file = read_file(("../models/" + uri).c_str()); }
                catch (const IoError &) { file = read_file((base_dir + uri).c_str()); }
            }
            p = file.data(); n = file.size();
        }
        DecodedImage img;
        if (looks_like_png(p, n)) img = decode_png(p, n);
        else if (looks_like_jpeg(p, n)) img = decode_jpeg(p, n);
        else throw UnsupportedError("image is neither PNG nor JPEG");
        HostTexture t;
        t.width = img.width; t.height = img.height; t.components = img.components;
        t.texels = std::move(img.texels);
        fresh.textures.push_back(std::move(t));
    }

    // loadMaterials (Scene.cu:59-86); tinygltf defaults: baseColorFactor [1,1,1,1], texture index -1, emissive 0
    const JsonValue &jmats = root.at("materials");
    for (size_t i = 0; i < jmats.size(); i++) {
        const JsonValue &pbr = jmats.at(i).at("pbrMetallicRoughness");
        drt_material m;
        std::memset(&m, 0, sizeof m);
        const JsonValue &col = pbr.at("baseColorFactor");
        for (int k = 0; k < 3; k++) m.albedo[k] = (float)(col.size() >= 3 ? col.at((size_t)k).as_double(1.0) : 1.0);
        const JsonValue &em = jmats.at(i).at("emissiveFactor");
        for (int k = 0; k < 3; k++) m.emissive[k] = (float)(em.size() >= 3 ? em.at((size_t)k).as_double(0.0) : 0.0);
        m.albedo_tex = (int32_t)pbr.at("baseColorTexture").at("index").as_int(-1);     // used as IMAGE index (Scene.cu:79)
        if (strict && m.albedo_tex >= 0) {                                             // spec: textures[index].source
            const JsonValue &jtex = root.at("textures");
            if ((size_t)m.albedo_tex >= jtex.size()) throw std::runtime_error("baseColorTexture.index out of range");
            m.albedo_tex = (int32_t)jtex.at((size_t)m.albedo_tex).at("source").as_int(-1);
        }
        m.roughness = (float)pbr.at("roughnessFactor").as_double(1.0);
        m.metallic = pbr.at("metallicFactor").as_double(1.0) > 0;
        m.transmission = 0;
        m.refractive_index = 1.45f;
        fresh.materials.push_back(m);
    }

    const JsonValue &jnodes = root.at("nodes");
    const JsonValue &jmeshes = root.at("meshes");
    if (strict) load_strict_geometry(root, buffers, views, fresh);
    // loadGLTFmodel (Scene.cu:192-314): every NODE contributes its mesh; transforms are ignored
    for (size_t n = 0; !strict && n < jnodes.size(); n++) {
        int64_t mesh_index = jnodes.at(n).at("mesh").as_int(-1);
        if (mesh_index < 0 || (size_t)mesh_index >= jmeshes.size())
            throw UnsupportedError("node without a mesh: the reference indexes meshes[-1] here (Scene.cu:199-200)");
        std::vector<float> pos, nrm, uv;
        std::vector<int32_t> mat;
        const JsonValue &prims = jmeshes.at((size_t)mesh_index).at("primitives");
        for (size_t p = 0; p < prims.size(); p++) {   // parseMesh (Scene.cu:120-178)
            const JsonValue &prim = prims.at(p);
            const JsonValue &attrs = prim.at("attributes");
            // a missing attribute silently maps to accessor 0 (std::map::operator[], Scene.cu:129-131)
            const BufferView &vp = accessor_view(attrs.at("POSITION").as_int(0));
            const BufferView &vn = accessor_view(attrs.at("NORMAL").as_int(0));
            const BufferView &vt = accessor_view(attrs.at("TEXCOORD_0").as_int(0));
            int64_t idx_accessor = prim.at("indices").as_int(-1);
            if (idx_accessor < 0) throw UnsupportedError("non-indexed primitive (the reference indexes accessors[-1])");
            const BufferView &vi = accessor_view(idx_accessor);
            const uint8_t *base = buffers[(size_t)vi.buffer].first;     // everything is read from the INDICES' buffer
            size_t base_len = buffers[(size_t)vi.buffer].second;
            int64_t i0 = vi.offset / 2, i1 = (vi.length + vi.offset) / 2;    // u16 units from byte 0 (Scene.cu:161,166)
            for (int64_t i = i0; i < i1; i++) {
                uint16_t index;
                std::memcpy(&index, base + 2 * (size_t)i, 2);
                size_t op = (size_t)vp.offset + 12u * index, on = (size_t)vn.offset + 12u * index, ot = (size_t)vt.offset + 8u * index;
                if (op + 12 > base_len || on + 12 > base_len || ot + 8 > base_len)
                    throw std::runtime_error("vertex index reads past the buffer");
                float f[3];
                std::memcpy(f, base + op, 12); pos.insert(pos.end(), f, f + 3);
                std::memcpy(f, base + on, 12); nrm.insert(nrm.end(), f, f + 3);
                std::memcpy(f, base + ot, 8);  uv.insert(uv.end(), f, f + 2);
            }
            int64_t count = jaccessors.at((size_t)idx_accessor).at("count").as_int(0);
            int32_t material = (int32_t)prim.at("material").as_int(-1);
            for (int64_t i = 0; i < count / 3; i++) mat.push_back(material);     // Scene.cu:172-175
        }
        size_t n_tris = pos.size() / 9;
        if (pos.size() % 9 != 0 || mat.size() < n_tris)
            throw UnsupportedError("index bufferView / accessor count mismatch (Scene.cu:301 would read out of bounds)");
        for (size_t t = 0; t < n_tris; t++) {
            int32_t m = mat[t];
            if (m < 0 || (size_t)m >= fresh.materials.size())
                throw UnsupportedError("primitive without a valid material (the kernel would index m_Material out of bounds)");
        }
        fresh.set_geometry(pos.data(), nrm.data(), uv.data(), mat.data(), (int32_t)n_tris);
    }
    for (const drt_material &m : fresh.materials)
        if (m.albedo_tex >= (int32_t)fresh.textures.size())
            throw UnsupportedError("baseColorTexture.index beyond the image list (used as image index, Scene.cu:79)");

    triangles = std::move(fresh.triangles);
    materials = std::move(fresh.materials);
    textures = std::move(fresh.textures);
    meshes = std::move(fresh.meshes);
    nodes.clear();
    revision = next_revision();
}

// ---------------------------------------------------------------------------------------------
// Binned-SAH builder (BVHBuilder.cu)
// ---------------------------------------------------------------------------------------------
namespace {

struct Extent { V3 lo, size; };

// getAbsoluteExtent (BVHBuilder.cuh:48-95): returns min and (max - min)
template <class Index>
Extent absolute_extent(const std::vector<drt_triangle> &tris, Index first, Index last, const int32_t *indirect) {
    V3 lo = v3(FLT_MAX, FLT_MAX, FLT_MAX), hi = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (Index i = first; i < last; i++) {
        const drt_triangle &t = tris[(size_t)(indirect ? indirect[i] : i)];
        for (int k = 0; k < 3; k++) {
            const float *p = t.vertex[k].position;
            lo.x = fminf(lo.x, p[0]); lo.y = fminf(lo.y, p[1]); lo.z = fminf(lo.z, p[2]);
            hi.x = fmaxf(hi.x, p[0]); hi.y = fmaxf(hi.y, p[1]); hi.z = fmaxf(hi.z, p[2]);
        }
    }
    return Extent{ lo, v3(hi.x - lo.x, hi.y - lo.y, hi.z - lo.z) };
}

drt_bvh_node fresh_node() {             // BVHNode.cuh:19-25, Bounds.cuh:12-13
    drt_bvh_node n;
    std::memset(&n, 0, sizeof n);
    n.bmin[0] = n.bmin[1] = n.bmin[2] = FLT_MAX;
    n.bmax[0] = n.bmax[1] = n.bmax[2] = -FLT_MAX;
    n.child1 = n.child2 = -1;
    n.prim_count = 0;
    n.prim_start = -1;
    return n;
}

void set_bounds(drt_bvh_node &n, const Extent &e) {      // Bounds3f(minextent, minextent + extent)
    st(n.bmin, e.lo);
    st(n.bmax, e.lo + e.size);
}

float surface_area(const float *lo, const float *hi) {   // Bounds.cu:4-10
    float planex = 2 * (hi[2] - lo[2]) * (hi[1] - lo[1]);
    float planey = 2 * (hi[2] - lo[2]) * (hi[0] - lo[0]);
    float planez = 2 * (hi[0] - lo[0]) * (hi[1] - lo[1]);
    return planex + planey + planez;
}

float node_area(const drt_bvh_node &n) { return n.prim_count == 0 ? 0.f : surface_area(n.bmin, n.bmax); }   // BVHNode.cuh:29-35

// The reference truncates a float SAH cost into an int (BVHBuilder.cu:284); it is an x86-64 build, where
// cvttss2si turns NaN / out-of-range into INT_MIN.
int truncate_like_x86(float f) {
    if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT_MIN;
    return (int)f;
}

struct Builder {
    std::vector<drt_triangle> &tris;
    int32_t bins;
    std::vector<int32_t> left_ids, right_ids;

    // binToShallowNodes (BVHBuilder.cu:216-255): classify, take bounds, no reordering
    void classify(drt_bvh_node &left, drt_bvh_node &right, float plane, int axis, int32_t first, int32_t last) {
        left_ids.clear(); right_ids.clear();
        for (int32_t i = first; i < last; i++)
            (tris[(size_t)i].centroid[axis] < plane ? left_ids : right_ids).push_back(i);
        left.prim_count = (int32_t)left_ids.size();
        set_bounds(left, absolute_extent<size_t>(tris, 0, left_ids.size(), left_ids.data()));
        right.prim_count = (int32_t)right_ids.size();
        set_bounds(right, absolute_extent<size_t>(tris, 0, right_ids.size(), right_ids.data()));
    }

    // binToNodes (BVHBuilder.cu:175-214).  std::partition on a vector iterator takes libstdc++'s
    // bidirectional algorithm; calling it here reproduces the reference's swap sequence and so the
    // in-leaf triangle order (which decides ties at equal hit distance).
    void split(drt_bvh_node &left, drt_bvh_node &right, float plane, int axis, int32_t first, int32_t last) {
        auto mid_it = std::partition(tris.begin() + first, tris.begin() + last,
                                     [plane, axis](const drt_triangle &t) { return t.centroid[axis] < plane; });
        int32_t mid = (int32_t)(mid_it - tris.begin());
        left.prim_start = first;
        left.prim_count = mid - first;
        set_bounds(left, absolute_extent<int32_t>(tris, left.prim_start, left.prim_start + left.prim_count, nullptr));
        right.prim_start = mid;
        right.prim_count = last - mid;
        set_bounds(right, absolute_extent<int32_t>(tris, right.prim_start, right.prim_start + right.prim_count, nullptr));
    }

    // makePartition (BVHBuilder.cu:257-346)
    void make_partition(int32_t first, int32_t last, drt_bvh_node &left_out, drt_bvh_node &right_out) {
        Extent ext = absolute_extent<int32_t>(tris, first, last, nullptr);
        drt_bvh_node parent = fresh_node();
        set_bounds(parent, ext);
        const float parent_area = surface_area(parent.bmin, parent.bmax);
        const float lo[3] = { ext.lo.x, ext.lo.y, ext.lo.z }, size[3] = { ext.size.x, ext.size.y, ext.size.z };
        drt_bvh_node left = fresh_node(), right = fresh_node();
        int lowest = INT_MAX, best_axis = 0;
        float best_plane = 0;
        for (int axis = 0; axis < 3; axis++) {
            float delta = size[axis] / bins;
            for (int i = 1; i < bins; i++) {
                float plane = lo[axis] + (i * delta);
                classify(left, right, plane, axis, first, last);
                // int cost = trav_cost + (SA_l / SA_p) * n_l * rayint_cost + (SA_r / SA_p) * n_r * rayint_cost
                float fcost = 1 + ((node_area(left) / parent_area) * left.prim_count * 2)
                                + ((node_area(right) / parent_area) * right.prim_count * 2);
                int cost = truncate_like_x86(fcost);
                if (cost < lowest) { lowest = cost; best_axis = axis; best_plane = plane; }   // strict <: first wins
            }
        }
        split(left_out, right_out, best_plane, best_axis, first, last);
    }
};

}  // namespace

void HostScene::build_bvh(int32_t target_leaf_prims, int32_t bin_count) {
    if (bin_count < 2) throw std::invalid_argument("bin_count must be >= 2");
    nodes.clear();
    revision = next_revision();
    const int32_t n = (int32_t)triangles.size();
    drt_bvh_node root = fresh_node();
    set_bounds(root, absolute_extent<int32_t>(triangles, 0, n, nullptr));
    root.prim_start = 0;
    root.prim_count = n;
    if (n <= target_leaf_prims) {                        // BVHBuilder.cu:34-43
        root.is_leaf = 1;
        nodes.push_back(root);
        return;
    }
    std::vector<drt_triangle> work = triangles;          // reordered copy, committed on success
    std::vector<drt_bvh_node> out;
    out.reserve((size_t)2 * (size_t)n + 2);
    Builder b{ work, bin_count, {}, {} };
    std::vector<int32_t> todo;                           // -1 = the detached root (appended last, BVHBuilder.cu:85)
    todo.push_back(-1);
    while (!todo.empty()) {
        int32_t cur = todo.back();
        todo.pop_back();
        drt_bvh_node node = cur < 0 ? root : out[(size_t)cur];
        if (node.prim_count <= target_leaf_prims) {      // BVHBuilder.cu:54-59
            node.is_leaf = 1;
        } else {
            drt_bvh_node l = fresh_node(), r = fresh_node();
            b.make_partition(node.prim_start, node.prim_start + node.prim_count, l, r);
            if (l.prim_count == 0 || r.prim_count == 0)
                throw BvhError("degenerate partition: every candidate plane leaves one side empty "
                               "(BVHBuilder.cu:49-83 never terminates on this input)");
            if (todo.size() + 2 > 512)                   // MAX_STACK_SIZE, BVHBuilder.cu:24
                throw BvhError("build stack deeper than the reference's 512-entry stack");
            out.push_back(l); node.child1 = (int32_t)out.size() - 1;
            out.push_back(r); node.child2 = (int32_t)out.size() - 1;
            todo.push_back(node.child1);                 // right child is built first (stack), BVHBuilder.cu:81-82
            todo.push_back(node.child2);
        }
        if (cur < 0) root = node; else out[(size_t)cur] = node;
    }
    out.push_back(root);
    triangles.swap(work);
    nodes.swap(out);
}

float HostScene::build_bvh_on_device(int32_t target_leaf_prims, int32_t bin_count, int device) {
    if (bin_count < 2) throw std::invalid_argument("bin_count must be >= 2");
    if ((int64_t)triangles.size() <= (int64_t)target_leaf_prims) {      // one leaf (BVHBuilder.cu:34-43): nothing to do in parallel
        build_bvh(target_leaf_prims, bin_count);
        return 0.f;
    }
    DeviceBuild built = drt::build_bvh_on_device(triangles, target_leaf_prims, bin_count, device);
    std::vector<drt_triangle> reordered(triangles.size());
    for (size_t i = 0; i < reordered.size(); i++) reordered[i] = triangles[built.order[i]];
    triangles.swap(reordered);
    nodes.swap(built.nodes);
    revision = next_revision();
    return built.device_ms;
}

int32_t HostScene::bvh_depth() const {
    if (nodes.empty()) return 0;
    int32_t depth = 0;
    std::vector<std::pair<int32_t, int32_t>> todo{ { (int32_t)nodes.size() - 1, 1 } };
    while (!todo.empty()) {
        auto [i, d] = todo.back();
        todo.pop_back();
        depth = std::max(depth, d);
        const drt_bvh_node &n = nodes[(size_t)i];
        if (!n.is_leaf) { todo.push_back({ n.child1, d + 1 }); todo.push_back({ n.child2, d + 1 }); }
    }
    return depth;
}

// ---------------------------------------------------------------------------------------------
// Device packing
// ---------------------------------------------------------------------------------------------
void HostScene::renumber_as_recursive_build() {
    if (nodes.size() < 2) return;
    const size_t n = nodes.size();
    std::vector<int32_t> order;                     // old indices in the order recursiveBuild appends them
    order.reserve(n);
    // iterative post-order: (node, state) -- state 0 = descend into child1, 1 = into child2, 2 = append both children
    std::vector<std::pair<int32_t, int>> st;
    st.emplace_back((int32_t)n - 1, 0);
    while (!st.empty()) {
        auto &top = st.back();
        const drt_bvh_node &nd = nodes[(size_t)top.first];
        if (nd.is_leaf) { st.pop_back(); continue; }
        if (top.second == 0) { top.second = 1; st.emplace_back(nd.child1, 0); }
        else if (top.second == 1) { top.second = 2; st.emplace_back(nd.child2, 0); }
        else { order.push_back(nd.child1); order.push_back(nd.child2); st.pop_back(); }
    }
    order.push_back((int32_t)n - 1);
    if (order.size() != n) throw BvhError("BVH is not a binary tree rooted at its last node");
    std::vector<int32_t> new_index(n, -1);
    for (size_t i = 0; i < n; i++) new_index[(size_t)order[i]] = (int32_t)i;
    std::vector<drt_bvh_node> out(n);
    for (size_t i = 0; i < n; i++) {
        drt_bvh_node nd = nodes[(size_t)order[i]];
        if (!nd.is_leaf) { nd.child1 = new_index[(size_t)nd.child1]; nd.child2 = new_index[(size_t)nd.child2]; }
        out[i] = nd;
    }
    if (!out.back().is_leaf) out.back().prim_start = -1;          // BVHBuilder.cu:112-118: only a leaf root gets primitive_start_idx = 0
    nodes.swap(out);
    revision = next_revision();
}

PackedScene HostScene::pack() const {
    if (nodes.empty()) throw std::invalid_argument("scene has no BVH: call build_bvh first (EditorLayer.cpp:52-55)");
    PackedScene ps;
    ps.depth = bvh_depth();
    const size_t n_nodes = nodes.size();
    std::vector<uint32_t> ref(n_nodes, kNoNode);
    for (size_t i = 0; i < n_nodes; i++) {
        const drt_bvh_node &n = nodes[i];
        if (n.is_leaf) {
            ref[i] = kLeafBit | (uint32_t)ps.leaves.size();
            ps.leaves.push_back(LeafRange{ n.prim_start, n.prim_count });
            ps.max_leaf = std::max(ps.max_leaf, n.prim_count);
        } else {
            ref[i] = (uint32_t)ps.inner.size();
            ps.inner.emplace_back();
        }
    }
    for (size_t i = 0; i < n_nodes; i++) {
        const drt_bvh_node &n = nodes[i];
        if (n.is_leaf) continue;
        InnerNode &rec = ps.inner[ref[i]];
        const drt_bvh_node &a = nodes[(size_t)n.child1], &b = nodes[(size_t)n.child2];
        std::memcpy(rec.c1min, a.bmin, 12); std::memcpy(rec.c1max, a.bmax, 12);
        std::memcpy(rec.c2min, b.bmin, 12); std::memcpy(rec.c2max, b.bmax, 12);
        rec.c1ref = ref[(size_t)n.child1];
        rec.c2ref = ref[(size_t)n.child2];
        rec._pad[0] = rec._pad[1] = 0;
    }
    const drt_bvh_node &root = nodes.back();             // root = last node (TraceRay.cu:20)
    ps.root_ref = ref[n_nodes - 1];
    std::memcpy(ps.root_min, root.bmin, 12);
    std::memcpy(ps.root_max, root.bmax, 12);

    // The kernels index mats[triangle.material], texs[material.albedo_tex] and the texel pool without checks: what the
    // programmatic API (drt_scene_set_geometry / _add_material) lets through must be refused here, before it is uploaded.
    for (size_t i = 0; i < n_nodes; i++) {
        const drt_bvh_node &n = nodes[i];
        if (n.is_leaf) {
            if (n.prim_start < 0 || n.prim_count < 0 || (size_t)n.prim_start + (size_t)n.prim_count > triangles.size())
                throw std::invalid_argument("BVH leaf " + std::to_string(i) + " points outside the triangle array");
        } else if (n.child1 < 0 || n.child2 < 0 || (size_t)n.child1 >= n_nodes || (size_t)n.child2 >= n_nodes)
            throw std::invalid_argument("BVH node " + std::to_string(i) + " has a child index outside the node array");
    }
    for (size_t i = 0; i < triangles.size(); i++)
        if (triangles[i].material < 0 || (size_t)triangles[i].material >= materials.size())
            throw std::invalid_argument("triangle " + std::to_string(i) + " uses material " + std::to_string(triangles[i].material) +
                                        " but the scene has " + std::to_string(materials.size()) + " materials");
    for (size_t i = 0; i < materials.size(); i++)
        if (materials[i].albedo_tex >= (int32_t)textures.size())
            throw std::invalid_argument("material " + std::to_string(i) + " uses texture " + std::to_string(materials[i].albedo_tex) +
                                        " but the scene has " + std::to_string(textures.size()) + " textures");
    ps.tri_hot.resize(triangles.size());
    ps.tri_cold.resize(triangles.size());
    for (size_t i = 0; i < triangles.size(); i++) {
        const drt_triangle &t = triangles[i];
        V3 p0 = ld(t.vertex[0].position);
        st(ps.tri_hot[i].v0, p0);
        st(ps.tri_hot[i].e1, ld(t.vertex[1].position) - p0);    // Intersection.cu:8
        st(ps.tri_hot[i].e2, ld(t.vertex[2].position) - p0);    // Intersection.cu:9
        std::memcpy(ps.tri_hot[i].fn, t.face_normal, 12);
        for (int k = 0; k < 3; k++) { ps.tri_cold[i].uv[k][0] = t.vertex[k].uv[0]; ps.tri_cold[i].uv[k][1] = t.vertex[k].uv[1]; }
        ps.tri_cold[i].material = t.material;
        ps.tri_cold[i]._pad = 0;
    }
    for (const drt_material &m : materials) {
        MatDev d;
        std::memcpy(d.albedo, m.albedo, 12);
        d.tex = m.albedo_tex;
        ps.mats.push_back(d);
        MatExt e;
        std::memset(&e, 0, sizeof e);
        std::memcpy(e.emissive, m.emissive, 12);
        e.roughness = m.roughness;
        e.metallic = m.metallic ? 1 : 0;
        e.transmission = m.transmission ? 1 : 0;
        e.refractive_index = m.refractive_index;
        ps.mats_ext.push_back(e);
    }
    for (const HostTexture &t : textures) {
        TexDev d;
        d.width = t.width; d.height = t.height; d.comps = t.components;
        d.offset = (uint32_t)ps.texels.size();
        ps.texels.insert(ps.texels.end(), t.texels.begin(), t.texels.end());
        ps.texels.resize(ps.texels.size() + (size_t)(t.width + 1) * (size_t)t.components, 0);
        while (ps.texels.size() % 16) ps.texels.push_back(0);
        ps.texs.push_back(d);
        if (t.components == 4) ps.any_alpha_texture = true;
    }
    return ps;
}

}  // namespace drt
