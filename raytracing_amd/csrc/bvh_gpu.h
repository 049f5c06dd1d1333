// Device-side BVH build (bvh_gpu.hip): Morton order + a hierarchy cut at the curve's cell boundaries, built level by level.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "bvh_build.h"

namespace mirt {

struct GpuBvh {
  float4* nodes = nullptr;          // device memory from hipMalloc, owned by the caller; layout as HostNode
  uint32_t n_nodes = 0;
  uint32_t root_ref = 0xFFFFFFFFu;  // REF_NONE when there are no primitives
  uint32_t height = 0;              // inner nodes on the longest root-to-leaf path
  std::vector<uint32_t> order;      // order[k] = index into the input of the primitive that is k-th in tree order
  float mn[3] = {0, 0, 0};
  float mx[3] = {0, 0, 0};
};

// Leaves hold up to `leaf_prims` (1..8) consecutive primitives of the tree order; the tree is at most
// `max_height` inner nodes tall.  Blocks until the build is done.
hipError_t build_bvh_gpu(const std::vector<BuildPrim>& prims, uint32_t leaf_prims, uint32_t max_height,
                         hipStream_t stream, GpuBvh& out);

}  // namespace mirt
