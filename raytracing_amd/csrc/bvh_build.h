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
  float lgrow;  // how the left child's box grows with the distance to the ray origin (BOX_GROW_SPHERE / the smallest
  float rmax[3];  // sphere radius below the child, 0 without spheres); filled in after the build (set_box_growth)
  float rgrow;
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

// Four-child node, 128 bytes = 8 quads, for trees walked from global memory: one fetch decides four children, so a
// ray takes fewer than half as many dependent fetches to reach its leaves.
//   q0..q2 = children's min x, min y, min z   q3..q5 = max x, max y, max z   q6 = child refs   q7 = unused
// A missing child has ref 0xFFFFFFFF and an inverted box.
struct WideNode {
  float lo[3][4];
  float hi[3][4];
  uint32_t child[4];
  uint32_t pad[4];
};
static_assert(sizeof(WideNode) == 128, "a wide node is two 64-byte lines");

struct WideBvh {
  std::vector<WideNode> nodes;
  uint32_t root_ref = 0xFFFFFFFFu;
  uint32_t height = 0;  // inner nodes on the longest root-to-leaf path
};

// Folds a two-child tree into four-child nodes: every node takes its two children and then, twice, opens the child
// with the largest surface area that is an inner node.  Leaves and their primitives stay as they are.
WideBvh widen_bvh(const std::vector<HostNode>& nodes, uint32_t root_ref);

// The four-child node as the device reads it: one 64-byte line.  The children's boxes are stored as 8-bit offsets
// on a grid that starts at the node's own minimum corner, with a spacing per axis of 1/255 of the node's extent
// (the next float up that passes the check below),
//   lo = fma(q_lo, spacing, origin)   hi = fma(q_hi, spacing, origin)      (one fused operation, rounded once),
// rounded outwards, so a decoded box always contains the child's true box: walks never miss, they only take the odd
// extra step.  Four 16-byte loads per step instead of seven; a walk from global memory pays for every load.
//   q0 = origin.xyz, spacing.x   q1 = lo.x[4] lo.y[4] lo.z[4] hi.x[4] as bytes
//   q2 = hi.y[4] hi.z[4] as bytes, spacing.y, spacing.z   q3 = child refs
struct QuantNode {
  float origin[3];
  float step_x;
  uint8_t lo[3][4];
  uint8_t hi[3][4];
  float step_y, step_z;
  uint32_t child[4];
};
static_assert(sizeof(QuantNode) == 64, "a quantised node is one 64-byte line");
std::vector<QuantNode> quantize_wide(const WideBvh& wide);

// Recomputes every child box of a built tree (same nodes, same leaves) from other primitive boxes: `prims` in tree
// order, as the build left them.  Used to fit one tree to the primitives' positions in several slices of time.
void refit_bvh(std::vector<HostNode>& nodes, uint32_t root_ref, const std::vector<BuildPrim>& prims);

// Fills in lgrow / rgrow of every node of a built tree (n_copies copies of n_nodes nodes, same tree): the largest
// of prim_growth over the primitives below the child; prim_growth in tree order, one value per primitive.
void set_box_growth(std::vector<HostNode>& nodes, uint32_t n_nodes, uint32_t n_copies, uint32_t root_ref,
                    const std::vector<float>& prim_growth);

}  // namespace mirt
