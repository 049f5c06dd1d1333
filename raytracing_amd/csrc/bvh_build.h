// Host builder of the flattened two-child BVH the traversal kernel walks.
// Replaces bvh.rs:24-81 (median split, one node per Vec entry) with a binned
// surface-area-heuristic build into 64-byte nodes that carry both child boxes.
#pragma once
#include <stdint.h>

#include <vector>

namespace mirt {

struct BuildPrim {
  float mn[3];
  float mx[3];
  uint32_t id;  // caller's primitive id; the build permutes the array into tree order
};

struct HostNode {  // 64 bytes, layout documented in scene_device.h
  float lmin[3];
  uint32_t left;
  float lmax[3];
  uint32_t right;
  float rmin[3];
  uint32_t pad0;
  float rmax[3];
  uint32_t pad1;
};
static_assert(sizeof(HostNode) == 64, "node must be one 64-byte line");

struct BuiltBvh {
  std::vector<HostNode> nodes;
  uint32_t root_ref = 0xFFFFFFFFu;  // REF_NONE when there are no primitives
  uint32_t height = 0;              // inner nodes on the longest root-to-leaf path
  float mn[3] = {0, 0, 0};
  float mx[3] = {0, 0, 0};
};

// Leaves hold up to `leaf_max` (<= 8) primitives that are consecutive in the permuted `prims`
// array: a leaf reference is 0x80000000 | (count - 1) << 28 | index of its first primitive.
// max_height: the traversal stack holds one entry per inner node on the
// current path, so the tree is kept no taller than this (falls back to
// balanced median splits when the budget gets tight).
// step_cost: what one more step down the tree costs, in primitive tests; above 1 the build keeps larger leaves
// (a smaller tree).
BuiltBvh build_bvh(std::vector<BuildPrim>& prims, uint32_t max_height, uint32_t leaf_max, float step_cost = 1.0f);

// Recomputes every child box of a built tree (same nodes, same leaves) from other primitive boxes: `prims` in tree
// order, as the build left them.  Used to fit one tree to the primitives' positions in several slices of time.
void refit_bvh(std::vector<HostNode>& nodes, uint32_t root_ref, const std::vector<BuildPrim>& prims);

}  // namespace mirt
