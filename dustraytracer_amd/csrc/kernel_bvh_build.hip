// kernel_bvh_build.hip -- the binned-SAH builder of BVHBuilder.cu:11-346 run on the GPU, level by level, making
// exactly the reference's decisions.
//
// What has to come out is what HostScene::build_bvh (scene_host.cpp) produces: the same BVHNode[] in the same order
// (children appended pairwise in the order the reference's explicit stack visits the nodes, root last) and the same
// triangle order inside every node (std::partition's swap sequence).  Everything the reference computes per node is
// either an integer count, a min/max over vertex positions (exact in any order) or a fixed scalar fp32 expression of
// those, so a parallel reduction gives the same bits as the serial loops:
//
//   level loop (host):   bin -> decide -> scatter -> swap        four launches per tree level, one 16-byte read-back
//   bin      one wave per chunk of <= 512 triangles of one node: for every candidate plane (3 axes x (bins-1),
//            BVHBuilder.cu:268-293) count the triangles with centroid < plane and take both sides' bounds
//            (binToShallowNodes, :216-255); wave reduction, then 13 atomics per candidate into the node's record
//   decide   one thread per node: the SAH cost of every candidate in the reference's order, truncated to int, first
//            lowest wins (:284-291); creates the two children, queues those with more than `leaf` triangles for the
//            next level and prefix-sums the chosen plane's per-chunk counts
//   scatter  std::partition (binToNodes, :185-194) is a Hoare scheme: the k-th element from the left that fails the
//            predicate is swapped with the k-th element from the right that passes it, until the two scans meet at
//            m = #passing.  With the prefix sums every element knows its k; the two lists are written out ...
//   swap     ... and thread k swaps the pair.  Ranges of different nodes are disjoint, so the order in which the
//            reference's stack visits nodes does not matter for the triangles.
//   finalize numbers the nodes the way the reference's stack does (a node's children are appended when the node is
//            popped, the right child is popped first, the root is appended last: BVHBuilder.cu:49-89) from the subtree
//            sizes, bottom-up then top-down, in one workgroup.
//
// The one freedom taken: a bound that is a zero takes the sign "-0 < +0" of an order-independent minimum, where the
// serial fminf/fmaxf loop keeps whichever zero came last.  No comparison in the renderer can tell the two apart.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cstring>
#include <string>

#include "bvh_build_device.hpp"
#include "scene_host.hpp"

namespace drt {
namespace {

constexpr int kTrisPerLane = 8;
constexpr int kChunk = 64 * kTrisPerLane;      // triangles one wave bins
constexpr int kAccWords = 13;                  // per candidate: n_left, left min[3], left max[3], right min[3], right max[3]
constexpr int kMaxLevels = 1024;

struct BuildNode {
    int32_t start, count;
    float lo[3], hi[3];                        // raw min / max of the vertex positions (getAbsoluteExtent, BVHBuilder.cuh:48-95)
    int32_t child_l, child_r;                  // build-node indices; -1 = leaf
};
struct ActiveNode { int32_t node, chunk_first, n_chunks; };
struct ChunkRef { int32_t slot, offset; };
struct Split { int32_t axis; float plane; int32_t n_left; int32_t n_swaps; };
struct Counters { uint32_t n_nodes, n_active_next, n_chunks_next, error; };

// order-preserving float <-> uint map, so that bounds can be merged with integer atomics (-0 sorts below +0)
__device__ __forceinline__ uint32_t float_key(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ float wave_min(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// fminf / fmaxf of the serial loop skip a NaN operand; so does this (and v_min_f32 / v_max_f32)
__device__ __forceinline__ float min_skip_nan(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float max_skip_nan(float a, float b) { return fmaxf(a, b); }

__device__ __forceinline__ float surface_area(const float *lo, const float *hi) {       // Bounds.cu:4-10
    const float planex = 2 * (hi[2] - lo[2]) * (hi[1] - lo[1]);
    const float planey = 2 * (hi[2] - lo[2]) * (hi[0] - lo[0]);
    const float planez = 2 * (hi[0] - lo[0]) * (hi[1] - lo[1]);
    return planex + planey + planez;
}
// area of Bounds3f(minextent, minextent + extent) as BVHNode::getSurfaceArea sees it (BVHNode.cuh:29-35)
__device__ __forceinline__ float area_of_extent(const float *lo, const float *hi, int32_t count) {
    if (count == 0) return 0.f;
    float bmax[3];
    for (int k = 0; k < 3; k++) bmax[k] = lo[k] + (hi[k] - lo[k]);
    return surface_area(lo, bmax);
}
__device__ __forceinline__ int truncate_like_x86(float f) {                               // int cost = float: cvttss2si
    if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT_MIN;
    return (int)f;
}

// ---- root: bounds of everything, the first node, its chunks ----
__global__ __launch_bounds__(256) void root_extent_kernel(const float4 *tri, uint32_t n, uint32_t *keys) {
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
        for (int k = 0; k < 3; k++) {
            const float4 v = tri[3 * (size_t)i + k];
            lo[0] = min_skip_nan(lo[0], v.x); lo[1] = min_skip_nan(lo[1], v.y); lo[2] = min_skip_nan(lo[2], v.z);
            hi[0] = max_skip_nan(hi[0], v.x); hi[1] = max_skip_nan(hi[1], v.y); hi[2] = max_skip_nan(hi[2], v.z);
        }
    for (int k = 0; k < 3; k++) { lo[k] = wave_min(lo[k]); hi[k] = wave_max(hi[k]); }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; k++) { atomicMin(&keys[k], float_key(lo[k])); atomicMax(&keys[3 + k], float_key(hi[k])); }
}

__global__ __launch_bounds__(256) void root_init_kernel(const uint32_t *keys, uint32_t n, BuildNode *nodes, ActiveNode *active,
                                                        ChunkRef *chunks, uint32_t *order, Counters *counters) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n_chunks = (n + kChunk - 1) / kChunk;
    if (i < n) order[i] = i;
    if (i < n_chunks) { chunks[i].slot = 0; chunks[i].offset = (int32_t)(i * kChunk); }
    if (i == 0) {
        BuildNode r;
        r.start = 0; r.count = (int32_t)n; r.child_l = r.child_r = -1;
        for (int k = 0; k < 3; k++) { r.lo[k] = key_float(keys[k]); r.hi[k] = key_float(keys[3 + k]); }
        nodes[0] = r;
        active[0].node = 0; active[0].chunk_first = 0; active[0].n_chunks = (int32_t)n_chunks;
        counters->n_nodes = 1; counters->n_active_next = 0; counters->n_chunks_next = 0; counters->error = 0;
    }
}

__global__ __launch_bounds__(256) void init_acc_kernel(uint32_t *acc, size_t words) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= words) return;
    const uint32_t w = (uint32_t)(i % kAccWords);
    // n_left = 0; minima start at FLT_MAX, maxima at -FLT_MAX (the empty Bounds3f of getAbsoluteExtent)
    acc[i] = w == 0 ? 0u : (((w - 1) / 3) % 2 == 0 ? 0xff7fffffu /* key(FLT_MAX) */ : 0x00800000u /* key(-FLT_MAX) */);
}

// ---- bin: one wave per chunk ----
struct LaneTris {
    float v[kTrisPerLane][9];
    float c[kTrisPerLane][3];
    bool valid[kTrisPerLane];
};

template <int AXIS>
__device__ __forceinline__ void bin_axis(const LaneTris &t, const BuildNode &node, int bins, uint32_t *acc_node, uint32_t *cnt_chunk,
                                         int lane) {
    const float lo = node.lo[AXIS], size = node.hi[AXIS] - node.lo[AXIS];
    const float delta = size / bins;                                   // BVHBuilder.cu:275
    for (int i = 1; i < bins; i++) {
        const float plane = lo + (i * delta);                          // :279
        uint32_t n_left = 0;
        float lmin[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, lmax[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
        float rmin[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, rmax[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
#pragma unroll
        for (int k = 0; k < kTrisPerLane; k++) {
            const bool left = t.valid[k] && t.c[k][AXIS] < plane;      // :230-239
            const bool right = t.valid[k] && !left;
            n_left += left ? 1u : 0u;
#pragma unroll
            for (int p = 0; p < 3; p++)
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    const float x = t.v[k][3 * p + a];
                    lmin[a] = min_skip_nan(lmin[a], left ? x : FLT_MAX);  lmax[a] = max_skip_nan(lmax[a], left ? x : -FLT_MAX);
                    rmin[a] = min_skip_nan(rmin[a], right ? x : FLT_MAX); rmax[a] = max_skip_nan(rmax[a], right ? x : -FLT_MAX);
                }
        }
        for (int o = 32; o > 0; o >>= 1) n_left += __shfl_xor(n_left, o);
#pragma unroll
        for (int a = 0; a < 3; a++) { lmin[a] = wave_min(lmin[a]); lmax[a] = wave_max(lmax[a]); rmin[a] = wave_min(rmin[a]); rmax[a] = wave_max(rmax[a]); }
        const int cand = AXIS * (bins - 1) + (i - 1);
        if (lane == 0) {
            uint32_t *rec = acc_node + (size_t)cand * kAccWords;
            cnt_chunk[cand] = n_left;
            atomicAdd(&rec[0], n_left);
#pragma unroll
            for (int a = 0; a < 3; a++) {
                atomicMin(&rec[1 + a], float_key(lmin[a])); atomicMax(&rec[4 + a], float_key(lmax[a]));
                atomicMin(&rec[7 + a], float_key(rmin[a])); atomicMax(&rec[10 + a], float_key(rmax[a]));
            }
        }
    }
}

__global__ __launch_bounds__(64) void bin_kernel(const float4 *tri, const uint32_t *order, const BuildNode *nodes, const ActiveNode *active,
                                                 const ChunkRef *chunks, int bins, uint32_t *acc, uint32_t *chunk_cnt) {
    const int lane = threadIdx.x;
    const ChunkRef ref = chunks[blockIdx.x];
    const BuildNode node = nodes[active[ref.slot].node];
    const int n_cand = 3 * (bins - 1);
    LaneTris t;
#pragma unroll
    for (int k = 0; k < kTrisPerLane; k++) {
        const int p = ref.offset + k * 64 + lane;
        t.valid[k] = p < node.count;
        const uint32_t id = t.valid[k] ? order[node.start + p] : order[node.start];
        const float4 a = tri[3 * (size_t)id], b = tri[3 * (size_t)id + 1], c = tri[3 * (size_t)id + 2];
        t.v[k][0] = a.x; t.v[k][1] = a.y; t.v[k][2] = a.z; t.v[k][3] = b.x; t.v[k][4] = b.y; t.v[k][5] = b.z;
        t.v[k][6] = c.x; t.v[k][7] = c.y; t.v[k][8] = c.z;
        t.c[k][0] = a.w; t.c[k][1] = b.w; t.c[k][2] = c.w;
    }
    uint32_t *acc_node = acc + (size_t)ref.slot * n_cand * kAccWords;
    uint32_t *cnt_chunk = chunk_cnt + (size_t)blockIdx.x * n_cand;
    bin_axis<0>(t, node, bins, acc_node, cnt_chunk, lane);
    bin_axis<1>(t, node, bins, acc_node, cnt_chunk, lane);
    bin_axis<2>(t, node, bins, acc_node, cnt_chunk, lane);
}

// ---- decide: one thread per node (makePartition, BVHBuilder.cu:257-346) ----
__global__ __launch_bounds__(64) void decide_kernel(BuildNode *nodes, const ActiveNode *active, uint32_t n_active, int bins, int leaf,
                                                    const uint32_t *acc, const uint32_t *chunk_cnt, uint32_t *chunk_prefix, Split *splits,
                                                    ActiveNode *active_next, ChunkRef *chunks_next, Counters *counters) {
    const uint32_t s = blockIdx.x * 64u + threadIdx.x;
    if (s >= n_active) return;
    const ActiveNode me = active[s];
    BuildNode node = nodes[me.node];
    const int n_cand = 3 * (bins - 1);
    const uint32_t *rec0 = acc + (size_t)s * n_cand * kAccWords;
    float pmax[3];
    for (int k = 0; k < 3; k++) pmax[k] = node.lo[k] + (node.hi[k] - node.lo[k]);
    const float parent_area = surface_area(node.lo, pmax);
    int lowest = INT_MAX, best = 0;
    float best_plane = 0;
    for (int axis = 0; axis < 3; axis++) {
        const float delta = (node.hi[axis] - node.lo[axis]) / bins;
        for (int i = 1; i < bins; i++) {
            const float plane = node.lo[axis] + (i * delta);
            const int cand = axis * (bins - 1) + (i - 1);
            const uint32_t *rec = rec0 + (size_t)cand * kAccWords;
            const int32_t n_l = (int32_t)rec[0], n_r = node.count - n_l;
            float llo[3], lhi[3], rlo[3], rhi[3];
            for (int k = 0; k < 3; k++) { llo[k] = key_float(rec[1 + k]); lhi[k] = key_float(rec[4 + k]); rlo[k] = key_float(rec[7 + k]); rhi[k] = key_float(rec[10 + k]); }
            // int cost = trav_cost + (SA_l / SA_p) * n_l * rayint_cost + (SA_r / SA_p) * n_r * rayint_cost   (:284)
            const float fcost = 1 + ((area_of_extent(llo, lhi, n_l) / parent_area) * n_l * 2)
                                  + ((area_of_extent(rlo, rhi, n_r) / parent_area) * n_r * 2);
            const int cost = truncate_like_x86(fcost);
            if (cost < lowest) { lowest = cost; best = cand; best_plane = plane; }       // strict <: the first lowest wins (:285-291)
        }
    }
    const uint32_t *rec = rec0 + (size_t)best * kAccWords;
    const int32_t n_l = (int32_t)rec[0], n_r = node.count - n_l;
    Split sp; sp.axis = best / (bins - 1); sp.plane = best_plane; sp.n_left = n_l; sp.n_swaps = 0;
    splits[s] = sp;
    uint32_t run = 0;
    for (int j = 0; j < me.n_chunks; j++) {
        chunk_prefix[me.chunk_first + j] = run;
        run += chunk_cnt[(size_t)(me.chunk_first + j) * n_cand + best];
    }
    if (n_l == 0 || n_r == 0) {                       // the reference never terminates on this input (:49-83)
        atomicOr(&counters->error, 1u);
        return;
    }
    const uint32_t base = atomicAdd(&counters->n_nodes, 2u);
    BuildNode child[2];
    child[0].start = node.start; child[0].count = n_l;
    child[1].start = node.start + n_l; child[1].count = n_r;
    for (int k = 0; k < 3; k++) {
        child[0].lo[k] = key_float(rec[1 + k]); child[0].hi[k] = key_float(rec[4 + k]);
        child[1].lo[k] = key_float(rec[7 + k]); child[1].hi[k] = key_float(rec[10 + k]);
    }
    for (int c = 0; c < 2; c++) {
        child[c].child_l = child[c].child_r = -1;
        nodes[base + c] = child[c];
        if (child[c].count > leaf) {                  // :54-59
            const uint32_t slot = atomicAdd(&counters->n_active_next, 1u);
            const uint32_t n_ch = ((uint32_t)child[c].count + kChunk - 1) / kChunk;
            const uint32_t c0 = atomicAdd(&counters->n_chunks_next, n_ch);
            active_next[slot].node = (int32_t)(base + c); active_next[slot].chunk_first = (int32_t)c0; active_next[slot].n_chunks = (int32_t)n_ch;
            for (uint32_t j = 0; j < n_ch; j++) { chunks_next[c0 + j].slot = (int32_t)slot; chunks_next[c0 + j].offset = (int32_t)(j * kChunk); }
        }
    }
    nodes[me.node].child_l = (int32_t)base;
    nodes[me.node].child_r = (int32_t)base + 1;
}

// ---- scatter + swap: std::partition's permutation ----
__global__ __launch_bounds__(64) void scatter_kernel(const float4 *tri, const uint32_t *order, const BuildNode *nodes, const ActiveNode *active,
                                                     const ChunkRef *chunks, const uint32_t *chunk_prefix, Split *splits,
                                                     uint32_t *false_left, uint32_t *true_right) {
    const int lane = threadIdx.x;
    const ChunkRef ref = chunks[blockIdx.x];
    const BuildNode node = nodes[active[ref.slot].node];
    const Split sp = splits[ref.slot];
    const int32_t m = sp.n_left;
    uint32_t running = chunk_prefix[blockIdx.x];
    uint32_t n_swaps = 0;
    for (int k = 0; k < kTrisPerLane; k++) {
        const int p = ref.offset + k * 64 + lane;
        const bool valid = p < node.count;
        bool pred = false;
        if (valid) {
            const uint32_t id = order[node.start + p];
            const float c = tri[3 * (size_t)id + sp.axis].w;
            pred = c < sp.plane;
        }
        const unsigned long long b = __ballot(pred);
        const uint32_t before = running + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));     // passing elements in [0, p)
        const bool misplaced_left = valid && p < m && !pred;
        if (misplaced_left) false_left[node.start + (p - (int32_t)before)] = (uint32_t)p;      // k = failing elements in [0, p)
        if (valid && p >= m && pred) true_right[node.start + (m - (int32_t)before - 1)] = (uint32_t)p;   // k = passing elements in (p, n)
        n_swaps += (uint32_t)__popcll(__ballot(misplaced_left));
        running += (uint32_t)__popcll(b);
    }
    if (lane == 0 && n_swaps) atomicAdd(&splits[ref.slot].n_swaps, (int32_t)n_swaps);
}

__global__ __launch_bounds__(64) void swap_kernel(uint32_t *order, const BuildNode *nodes, const ActiveNode *active, const ChunkRef *chunks,
                                                  const Split *splits, const uint32_t *false_left, const uint32_t *true_right) {
    const int lane = threadIdx.x;
    const ChunkRef ref = chunks[blockIdx.x];
    const int32_t start = nodes[active[ref.slot].node].start;
    const int32_t n_swaps = splits[ref.slot].n_swaps;
    for (int k = 0; k < kTrisPerLane; k++) {
        const int i = ref.offset + k * 64 + lane;
        if (i < n_swaps) {
            const uint32_t a = (uint32_t)start + false_left[start + i], b = (uint32_t)start + true_right[start + i];
            const uint32_t ta = order[a], tb = order[b];
            order[a] = tb; order[b] = ta;
        }
    }
}

// ---- finalize: the reference's node numbering ----
__global__ __launch_bounds__(1024) void finalize_kernel(const BuildNode *nodes, uint32_t n_nodes, const uint32_t *level_begin, int n_levels,
                                                        int leaf, uint32_t *internal, uint32_t *pair_base, uint32_t *pending, uint32_t *final_index,
                                                        drt_bvh_node *out, Counters *counters) {
    const uint32_t tid = threadIdx.x;
    // subtree sizes, deepest level first: internal[x] = nodes of x's subtree that get split
    for (int l = n_levels - 1; l >= 0; l--) {
        for (uint32_t x = level_begin[l] + tid; x < level_begin[l + 1]; x += 1024u)
            internal[x] = nodes[x].child_l < 0 ? 0u : 1u + internal[nodes[x].child_l] + internal[nodes[x].child_r];
        __syncthreads();
    }
    if (tid == 0) { pair_base[0] = 0; pending[0] = 0; final_index[0] = n_nodes - 1; }     // the root is appended last (BVHBuilder.cu:85)
    __syncthreads();
    // A node popped from the stack appends its two children (child1 then child2), then the stack pops child2 first: the whole
    // right subtree is numbered before anything of the left one (:76-82).
    for (int l = 0; l < n_levels; l++) {
        for (uint32_t x = level_begin[l] + tid; x < level_begin[l + 1]; x += 1024u) {
            const int32_t cl = nodes[x].child_l, cr = nodes[x].child_r;
            if (cl < 0) continue;
            const uint32_t base = pair_base[x];
            final_index[cl] = base; final_index[cr] = base + 1;
            pair_base[cr] = base + 2;
            pair_base[cl] = base + 2 + 2 * internal[cr];
            pending[cr] = pending[x] + 1;              // the left sibling waits on the stack while the right subtree is built
            pending[cl] = pending[x];
            if (pending[x] + 2 > 512) atomicOr(&counters->error, 2u);                      // MAX_STACK_SIZE, BVHBuilder.cu:24
        }
        __syncthreads();
    }
    for (uint32_t x = tid; x < n_nodes; x += 1024u) {
        const BuildNode n = nodes[x];
        drt_bvh_node o;
        memset(&o, 0, sizeof o);
        o.is_leaf = n.child_l < 0 ? 1 : 0;
        for (int k = 0; k < 3; k++) { o.bmin[k] = n.lo[k]; o.bmax[k] = n.lo[k] + (n.hi[k] - n.lo[k]); }     // Bounds3f(minextent, minextent + extent)
        o.child1 = n.child_l < 0 ? -1 : (int32_t)final_index[n.child_l];
        o.child2 = n.child_r < 0 ? -1 : (int32_t)final_index[n.child_r];
        o.prim_count = n.count;
        o.prim_start = n.start;
        out[final_index[x]] = o;
    }
}

struct Buffers {
    std::vector<void *> owned;
    ~Buffers() { for (void *p : owned) (void)hipFree(p); }
    template <class T> T *alloc(size_t n) {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T));
        if (e != hipSuccess) throw DeviceError(std::string("hipMalloc (BVH build): ") + hipGetErrorString(e));
        owned.push_back(p);
        return static_cast<T *>(p);
    }
    void release(void *p) {
        auto it = std::find(owned.begin(), owned.end(), p);
        if (it != owned.end()) { (void)hipFree(p); owned.erase(it); }
    }
};

void check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw DeviceError(std::string(what) + ": " + hipGetErrorString(e));
}

}  // namespace

DeviceBuild build_bvh_on_device(const std::vector<drt_triangle> &tris, int32_t leaf, int32_t bins, int device) {
    if (bins < 2) throw std::invalid_argument("bin_count must be >= 2");
    if (tris.empty() || (int64_t)tris.size() <= (int64_t)leaf) throw std::invalid_argument("nothing to split: the host builder handles a one-node tree");
    if (tris.size() > (size_t)INT32_MAX / 2) throw std::invalid_argument("too many triangles");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) throw DeviceError("no usable HIP device: this library has no CPU fallback");
    if (device < 0 || device >= n_dev) throw std::invalid_argument("device index out of range");
    check(hipSetDevice(device), "hipSetDevice");

    const uint32_t n = (uint32_t)tris.size();
    const int n_cand = 3 * (bins - 1);
    // vertex positions + centroid, 48 B per triangle: {v0, c.x} {v1, c.y} {v2, c.z}
    std::vector<float> packed((size_t)n * 12);
    for (uint32_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            std::memcpy(&packed[(size_t)i * 12 + 4 * k], tris[i].vertex[k].position, 12);
            packed[(size_t)i * 12 + 4 * k + 3] = tris[i].centroid[k];
        }

    Buffers buf;
    float4 *d_tri = buf.alloc<float4>((size_t)n * 3);
    uint32_t *d_order = buf.alloc<uint32_t>(n);
    BuildNode *d_nodes = buf.alloc<BuildNode>((size_t)2 * n);
    const size_t max_active = (size_t)n / (size_t)(std::max(leaf, 0) + 1) + 2;
    const size_t max_chunks = (size_t)n / kChunk + max_active + 2;
    ActiveNode *d_active[2] = { buf.alloc<ActiveNode>(max_active), buf.alloc<ActiveNode>(max_active) };
    ChunkRef *d_chunks[2] = { buf.alloc<ChunkRef>(max_chunks), buf.alloc<ChunkRef>(max_chunks) };
    Split *d_splits = buf.alloc<Split>(max_active);
    uint32_t *d_chunk_prefix = buf.alloc<uint32_t>(max_chunks);
    uint32_t *d_false_left = buf.alloc<uint32_t>(n), *d_true_right = buf.alloc<uint32_t>(n);
    uint32_t *d_keys = buf.alloc<uint32_t>(6);
    Counters *d_counters = buf.alloc<Counters>(1);
    uint32_t *d_acc = nullptr, *d_chunk_cnt = nullptr;
    size_t acc_words = 0, cnt_words = 0;

    hipStream_t stream = nullptr;
    hipEvent_t ev0, ev1;
    check(hipEventCreate(&ev0), "hipEventCreate");
    check(hipEventCreate(&ev1), "hipEventCreate");
    struct EventGuard { hipEvent_t a, b; ~EventGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } guard{ ev0, ev1 };

    check(hipMemcpy(d_tri, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy (triangles)");
    check(hipEventRecord(ev0, stream), "hipEventRecord");
    const uint32_t key_init[6] = { 0xff7fffffu, 0xff7fffffu, 0xff7fffffu, 0x00800000u, 0x00800000u, 0x00800000u };
    check(hipMemcpyAsync(d_keys, key_init, sizeof key_init, hipMemcpyHostToDevice, stream), "hipMemcpy (keys)");
    hipLaunchKernelGGL(root_extent_kernel, dim3(std::min<uint32_t>((n + 255) / 256, 2048)), dim3(256), 0, stream, d_tri, n, d_keys);
    hipLaunchKernelGGL(root_init_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_keys, n, d_nodes, d_active[0], d_chunks[0], d_order, d_counters);

    std::vector<uint32_t> level_begin{ 0, 1 };
    uint32_t n_active = 1, n_chunks = (n + kChunk - 1) / kChunk;
    int cur = 0;
    uint32_t error = 0;
    while (n_active > 0) {
        if ((int)level_begin.size() > kMaxLevels) throw BvhError("tree deeper than 1024 levels");
        const size_t need_acc = (size_t)n_active * n_cand * kAccWords, need_cnt = (size_t)n_chunks * n_cand;
        if (need_acc > acc_words) { if (d_acc) buf.release(d_acc); acc_words = need_acc + need_acc / 2; d_acc = buf.alloc<uint32_t>(acc_words); }
        if (need_cnt > cnt_words) { if (d_chunk_cnt) buf.release(d_chunk_cnt); cnt_words = need_cnt + need_cnt / 2; d_chunk_cnt = buf.alloc<uint32_t>(cnt_words); }
        hipLaunchKernelGGL(init_acc_kernel, dim3((unsigned)((need_acc + 255) / 256)), dim3(256), 0, stream, d_acc, need_acc);
        hipLaunchKernelGGL(bin_kernel, dim3(n_chunks), dim3(64), 0, stream, d_tri, d_order, d_nodes, d_active[cur], d_chunks[cur], bins, d_acc, d_chunk_cnt);
        hipLaunchKernelGGL(decide_kernel, dim3((n_active + 63) / 64), dim3(64), 0, stream, d_nodes, d_active[cur], n_active, bins, leaf, d_acc,
                           d_chunk_cnt, d_chunk_prefix, d_splits, d_active[cur ^ 1], d_chunks[cur ^ 1], d_counters);
        hipLaunchKernelGGL(scatter_kernel, dim3(n_chunks), dim3(64), 0, stream, d_tri, d_order, d_nodes, d_active[cur], d_chunks[cur], d_chunk_prefix,
                           d_splits, d_false_left, d_true_right);
        hipLaunchKernelGGL(swap_kernel, dim3(n_chunks), dim3(64), 0, stream, d_order, d_nodes, d_active[cur], d_chunks[cur], d_splits, d_false_left,
                           d_true_right);
        Counters c;
        check(hipMemcpyAsync(&c, d_counters, sizeof c, hipMemcpyDeviceToHost, stream), "hipMemcpy (counters)");
        check(hipStreamSynchronize(stream), "BVH build level");
        error |= c.error;
        if (error) break;
        level_begin.push_back(c.n_nodes);
        n_active = c.n_active_next; n_chunks = c.n_chunks_next;
        if (n_active > max_active || n_chunks > max_chunks) throw BvhError("internal: level larger than its bound");
        check(hipMemsetAsync(&d_counters->n_active_next, 0, 2 * sizeof(uint32_t), stream), "hipMemset (counters)");   // stream-ordered: no second sync
        cur ^= 1;
    }
    if (error & 1u)
        throw BvhError("degenerate partition: every candidate plane leaves one side empty (BVHBuilder.cu:49-83 never terminates on this input)");

    const uint32_t n_nodes = level_begin.back();
    const int n_levels = (int)level_begin.size() - 1;
    uint32_t *d_level_begin = buf.alloc<uint32_t>(level_begin.size());
    uint32_t *d_internal = buf.alloc<uint32_t>(n_nodes), *d_pair = buf.alloc<uint32_t>(n_nodes), *d_pending = buf.alloc<uint32_t>(n_nodes),
             *d_final = buf.alloc<uint32_t>(n_nodes);
    drt_bvh_node *d_out = buf.alloc<drt_bvh_node>(n_nodes);
    check(hipMemcpyAsync(d_level_begin, level_begin.data(), level_begin.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream), "hipMemcpy (levels)");
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(1024), 0, stream, d_nodes, n_nodes, d_level_begin, n_levels, leaf, d_internal, d_pair, d_pending,
                       d_final, d_out, d_counters);
    check(hipEventRecord(ev1, stream), "hipEventRecord");
    DeviceBuild out;
    out.nodes.resize(n_nodes);
    out.order.resize(n);
    Counters c;
    check(hipMemcpy(out.nodes.data(), d_out, (size_t)n_nodes * sizeof(drt_bvh_node), hipMemcpyDeviceToHost), "hipMemcpy (nodes)");
    check(hipMemcpy(out.order.data(), d_order, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost), "hipMemcpy (order)");
    check(hipMemcpy(&c, d_counters, sizeof c, hipMemcpyDeviceToHost), "hipMemcpy (counters)");
    check(hipGetLastError(), "BVH build kernels");
    if (c.error & 2u) throw BvhError("build stack deeper than the reference's 512-entry stack");
    check(hipEventElapsedTime(&out.device_ms, ev0, ev1), "hipEventElapsedTime");
    out.levels = n_levels;
    return out;
}

}  // namespace drt
