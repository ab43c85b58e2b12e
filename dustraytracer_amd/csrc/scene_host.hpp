// scene_host.hpp -- host-side scene: glTF flattening, binned-SAH BVH build, packing for the device.
// Replaces struct Scene (Core/Scene/Scene.cuh:41-57) and class BVHBuilder (Core/BVH/BVHBuilder.cuh:12-96).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/drt.h"
#include "device_scene.hpp"

namespace drt {

struct HostTexture {
    int width = 0, height = 0, components = 0;
    std::vector<uint8_t> texels;
};

// 3-float helpers with one IEEE rounding per operation, evaluated in the order the reference's
// helper_math.cuh operators evaluate them (host build: -ffp-contract=off, no FMA).
struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { return V3{ x, y, z }; }
inline V3 operator+(V3 a, V3 b) { return V3{ a.x + b.x, a.y + b.y, a.z + b.z }; }
inline V3 operator-(V3 a, V3 b) { return V3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
inline V3 operator*(V3 a, float s) { return V3{ a.x * s, a.y * s, a.z * s }; }
inline V3 operator*(float s, V3 a) { return V3{ s * a.x, s * a.y, s * a.z }; }
inline V3 operator/(V3 a, float s) { return V3{ a.x / s, a.y / s, a.z / s }; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return V3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
V3 normalize(V3 v);   // v * (1.0f / sqrtf(dot(v, v)))  (helper_math.cuh:1325-1328 with :78-81)

class HostScene {
public:
    std::vector<drt_triangle> triangles;     // m_PrimitivesBuffer (reordered in place by build_bvh)
    std::vector<drt_material> materials;     // m_Material
    std::vector<HostTexture> textures;       // m_Textures
    std::vector<drt_mesh> meshes;            // m_Meshes
    std::vector<drt_bvh_node> nodes;         // m_BVHNodes, root last
    uint64_t revision = next_revision();     // process-wide unique stamp, renewed on every change; renderers re-upload when it moves
    static uint64_t next_revision();

    void clear();
    // strict = false: the reference's reading of the file, quirks included (Scene.cu); true: the glTF 2.0 specification's
    void load_gltf(const char *path, bool strict = false);        // throws std::runtime_error / UnsupportedError
    void set_geometry(const float *pos, const float *nrm, const float *uv, const int32_t *mat, int32_t n_tris);
    void build_bvh(int32_t target_leaf_prims, int32_t bin_count);   // throws BvhError when the reference would hang
    // BVHBuilder::build (BVHBuilder.cu:100-173): the same partitions, hence the same tree and triangle order, with the node
    // array in the recursion's order (a node's two children follow both of their subtrees; root last; a split root keeps
    // primitive_start_idx = -1).  Call after build_bvh / build_bvh_on_device.
    void renumber_as_recursive_build();
    // The same tree built on a GPU (kernel_bvh_build.hip); returns the device time in ms.  Throws DeviceError without one.
    float build_bvh_on_device(int32_t target_leaf_prims, int32_t bin_count, int device);
    int32_t bvh_depth() const;

    // Device-layout image of the scene (see device_scene.hpp); throws when there is no BVH.
    PackedScene pack() const;
};

struct UnsupportedError : std::runtime_error { using std::runtime_error::runtime_error; };
struct IoError : std::runtime_error { using std::runtime_error::runtime_error; };
struct BvhError : std::runtime_error { using std::runtime_error::runtime_error; };

}  // namespace drt
