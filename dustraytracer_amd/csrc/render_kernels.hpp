// render_kernels.hpp -- launch wrappers of render_kernels.hip (the seam that replaces InvokeRenderKernel,
// Core/Kernel/RenderKernel.cuh:12-14; grid/block shape is an internal choice here).
#pragma once
#include <hip/hip_runtime.h>

#include "device_scene.hpp"

namespace drt {

hipError_t launch_render(const SceneView &scene, const FrameParams &frame, int bvh_depth, bool count_work,
                         hipStream_t stream, const char **kernel_name);

hipError_t launch_assemble(const void *gathered, void *image, uint32_t width, uint32_t height, uint32_t stripe_rows,
                           uint32_t world, uint32_t padded_rows, hipStream_t stream);

}  // namespace drt
