// bvh_build_device.hpp -- BVHBuilder::buildIterative (Core/BVH/BVHBuilder.cu:11-346) on the GPU: same nodes, same node
// order, same triangle order as HostScene::build_bvh (kernel_bvh_build.hip).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <vector>

#include "../../include/drt.h"

namespace drt {

struct DeviceError : std::runtime_error { using std::runtime_error::runtime_error; };

struct DeviceBuild {
    std::vector<drt_bvh_node> nodes;      // reference order, root last
    std::vector<uint32_t> order;          // triangle that ends up at position i
    float device_ms = 0;                  // first kernel to last, including the per-level read-backs
    int levels = 0;
};

// Throws std::invalid_argument, BvhError (degenerate input, as the host builder) or DeviceError.  There is no CPU path
// behind this call: without a HIP device it fails.
DeviceBuild build_bvh_on_device(const std::vector<drt_triangle> &tris, int32_t target_leaf_prims, int32_t bin_count, int device);

}  // namespace drt
