// device_scene.hpp -- how the scene and the per-call constants are laid out in HBM / LDS.
//
// The reference keeps AoS structs (BVHNode 44 B, Triangle 128 B, Material 44 B) and chases pointers
// (Core/BVH/BVHTraversal.cuh:34,48,60-61).  Here the same information is split by access pattern:
//
//   InnerNode  64 B  both child AABBs + both child references ("child-box-pair record"): one interior visit
//                    = 4 x 16-byte loads, and a child's kind (leaf / interior) is known without touching it
//   LeafRange   8 B  [start, count) into the triangle arrays
//   TriHot     48 B  v0, e1 = v1-v0, e2 = v2-v0 (what Intersection.cu:8-9 recomputes per test -- the two
//                    subtractions are exact IEEE ops, so precomputing them is bit-identical) + face normal
//   TriCold    32 B  the three UVs + material id: read only for the closest hit / alpha test
//   MatDev     16 B  albedo + albedo texture index (all the kernel reads, RayGen.cuh:112-117)
//   TexDev     16 B  width, height, channels, byte offset into one texel pool
//
// Small scenes (everything but the texel pool <= kLdsSceneBudget) are copied into LDS by every workgroup.
#pragma once
#include <cstdint>
#include <vector>

namespace drt {

constexpr uint32_t kLeafBit = 0x80000000u;     // node reference: bit 31 -> leaf id, else interior record index
constexpr uint32_t kNoNode = 0xFFFFFFFFu;      // empty scene

struct alignas(16) InnerNode {
    float c1min[3], c1max[3];
    float c2min[3], c2max[3];
    uint32_t c1ref, c2ref;
    uint32_t _pad[2];
};
static_assert(sizeof(InnerNode) == 64, "InnerNode layout");

struct LeafRange { int32_t start, count; };

struct alignas(16) TriHot { float v0[3], e1[3], e2[3], fn[3]; };
static_assert(sizeof(TriHot) == 48, "TriHot layout");

struct alignas(16) TriCold { float uv[3][2]; int32_t material; int32_t _pad; };
static_assert(sizeof(TriCold) == 32, "TriCold layout");

struct alignas(16) MatDev { float albedo[3]; int32_t tex; };
struct alignas(16) TexDev { int32_t width, height, comps; uint32_t offset; };
// Material fields the reference loads and never reads (Material.cuh:9,17,20; Scene.cu:71-75): only the opt-in material model
// (drt_renderer_set_material_model) reads them, and only the general kernel
struct alignas(16) MatExt { float emissive[3]; float roughness; int32_t metallic; int32_t transmission; float refractive_index; int32_t _pad; };

// Host-side image of the device buffers.
struct PackedScene {
    std::vector<InnerNode> inner;
    std::vector<LeafRange> leaves;
    std::vector<TriHot> tri_hot;
    std::vector<TriCold> tri_cold;
    std::vector<MatDev> mats;
    std::vector<MatExt> mats_ext;
    std::vector<TexDev> texs;
    std::vector<uint8_t> texels;     // every texture followed by (width+1) zero texels (latent OOB read of Texture.cu:35-49)
    uint32_t root_ref = kNoNode;
    float root_min[3] = { 0, 0, 0 }, root_max[3] = { 0, 0, 0 };
    int32_t depth = 0;               // BVH levels; traversal stack never holds more than depth entries
    int32_t max_leaf = 0;
    bool any_alpha_texture = false;  // some texture has 4 channels -> AnyHit may reject hits
};

// What the kernels receive (by value, as one kernel argument).
struct SceneView {
    const InnerNode *inner;
    const LeafRange *leaves;
    const TriHot *tri_hot;
    const TriCold *tri_cold;
    const MatDev *mats;
    const MatExt *mats_ext;
    const TexDev *texs;
    const uint8_t *texels;
    uint32_t n_inner, n_leaves, n_tris, n_mats, n_texs;
    uint32_t root_ref;
    float root_min[3], root_max[3];
};

// Per-call constants.  Everything that the reference recomputes per pixel from Camera / RendererSettings
// but that is constant over the frame is hoisted to the host (Camera.cu:84-103, RayGen.cuh:68-72).
struct FrameParams {
    // Camera::GetRay (Camera.cu:82-123)
    float cam_pos[3];
    float fwd_focus[3];        // normalize(forward) * focus_dist
    float horizontal[3];       // world_image_plane_width  * right
    float vertical[3];         // world_image_plane_height * up
    float disk_u[3], disk_v[3];
    int32_t defocus;           // !(defocus_angle <= 0)
    float exposure;
    // RayGen (RayGen.cuh:68-72)
    float sunpos[3], suncol[3];
    float sky_color[3];
    float sky_intensity;
    int32_t gamma_correction, tone_mapping, enable_sunlight;
    int32_t bounce_limit;
    int32_t render_mode, debug_mode;
    // RenderKernel.cu:20-35
    uint32_t width, height;          // full image
    uint32_t frame_first, n_frames;  // frame indices frame_first .. frame_first+n_frames-1 (>= 1)
    // sharding: this device owns rows y with (y / stripe_rows) % world == rank, stored compactly
    uint32_t stripe_rows, rank, world, local_rows;
    float *accum;                    // float3[width*local_rows]
    float *rgba;                     // float4[width*local_rows]
    unsigned long long *counters;    // drt_counters as 24 x u64, or nullptr
    unsigned long long *span;        // wave_queue: {max(~start), max(end)} of the kernel on the constant-rate wall clock, or nullptr
    // wave_queue phase voting: a phase other than T runs as soon as this many lanes wait for it
    int32_t vote_node, vote_shade, vote_dir, vote_spec;
    int32_t frames_in_flight;         // launches the caller keeps in flight on this device (grid sizing of small launches)
    int32_t vote_tail_node, vote_tail_shade;   // thresholds once the sample queue is empty (drain of the launch)
    int32_t leaf_chain;                        // T steps take the next leaf off the stack themselves (shallow trees)
    uint32_t row_step;                         // wave_queue work order: stride over the tile rows, coprime to their number
    // opt-in material model (not reference behaviour; all zero = the reference's image): see include/drt.h drt_material_model
    int32_t ext_emissive, ext_specular;
    float ext_emissive_scale;
    int32_t ext_transmission;
    int32_t inline_resolve;                    // the launch holds ONE frame: the tracing kernel adds each sample to the running sum and
                                               // writes the resolved texel itself (RenderKernel.cu:29-34), no sample buffer, no resolve kernel
};

}  // namespace drt
