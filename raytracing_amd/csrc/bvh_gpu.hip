// Device-side BVH build for scenes whose host build (bvh_build.cpp) would take seconds to minutes.
//
// Replaces bvh.rs:24-81 (recursive sort + median split, single-threaded, O(N log^2 N)) for large N:
//   1. 63-bit Morton code of every primitive's box centre (21 bits per axis, cubic cells over the centroid bounds),
//   2. radix sort of (code, primitive number): the tree order of the primitives,
//   3. the hierarchy top down, one launch per level: a range of the sorted order is cut where the highest
//      differing bit of its codes flips (the cell boundary of the space-filling curve), or in the middle when
//      the codes are equal or when the height left under the traversal stack's depth does not allow an uneven
//      cut; ranges of at most `leaf_prims` primitives are leaves,
//   4. the boxes bottom up, again one launch per level, no atomics,
//   5. the few primitives that span a large part of the scene (a ground sphere under a field of marbles) are
//      kept out of 1-4 -- inside the hierarchy each would blow up the box of every node above it -- and hang
//      off a short chain of nodes above the root instead, where they are tested first.
// The closest-hit result does not depend on the tree's shape, only the time to find it.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "bvh_gpu.h"

namespace mirt {

namespace {

constexpr int kThreads = 256;

struct DBox {
  float4 mn, mx;  // .w unused
};

__device__ __forceinline__ uint64_t spread21(uint32_t v) {  // ...c b a -> ...00c00b00a
  uint64_t x = v & 0x1FFFFFu;
  x = (x | (x << 32)) & 0x001F00000000FFFFull;
  x = (x | (x << 16)) & 0x001F0000FF0000FFull;
  x = (x | (x << 8)) & 0x100F00F00F00F00Full;
  x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}

__global__ void k_morton(const BuildPrim* __restrict__ prims, uint32_t n, float3 origin, float3 scale,
                         uint64_t* __restrict__ keys, uint32_t* __restrict__ values) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const BuildPrim p = prims[i];
  const float cx = 0.5f * (p.mn[0] + p.mx[0]), cy = 0.5f * (p.mn[1] + p.mx[1]), cz = 0.5f * (p.mn[2] + p.mx[2]);
  auto cell = [](float c, float o, float s) {
    const float f = (c - o) * s;  // in [0, 2^21) for finite input
    return (uint32_t)fminf(fmaxf(f, 0.0f), 2097151.0f);
  };
  keys[i] = spread21(cell(cx, origin.x, scale.x)) | (spread21(cell(cy, origin.y, scale.y)) << 1) |
            (spread21(cell(cz, origin.z, scale.z)) << 2);
  values[i] = i;
}

// A range [a, b) of the sorted order waiting to become a node or a leaf, and where its reference goes.
struct Range {
  uint32_t a, b;
  uint32_t parent;  // node whose child this is, NO_PARENT for the root
  uint32_t side;    // 0 left, 1 right
};
constexpr uint32_t NO_PARENT = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t leaf_ref(uint32_t a, uint32_t b) { return 0x80000000u | ((b - a - 1u) << 28) | a; }

// Decides for every range of a level whether it becomes a node (flag 1) and where that node cuts it.
// `budget`: inner nodes this level's ranges may still stack on top of each other.
__global__ void k_split(const Range* __restrict__ ranges, uint32_t n_ranges, const uint64_t* __restrict__ keys,
                        uint32_t leaf_prims, uint32_t budget, uint32_t* __restrict__ flags, uint32_t* __restrict__ cuts) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_ranges) return;
  const uint32_t a = ranges[i].a, b = ranges[i].b, count = b - a;
  if (count <= leaf_prims) { flags[i] = 0; return; }
  flags[i] = 1;
  uint32_t cut = a + (count + 1u) / 2u;
  const uint64_t first = keys[a], last = keys[b - 1];
  const uint32_t need = 32u - (uint32_t)__clz((int)(count - 1u));  // levels a halving build of `count` needs at most
  if (first != last && need + 1u < budget) {
    // the last position whose code still shares more leading bits with the first one than the whole range does
    const int prefix = __clzll((long long)(first ^ last));
    uint32_t s = a, step = b - 1u - a;
    do {
      step = (step + 1u) >> 1;
      const uint32_t t = s + step;
      if (t < b - 1u && __clzll((long long)(first ^ keys[t])) > prefix) s = t;
    } while (step > 1u);
    cut = s + 1u;
  }
  cuts[i] = cut;
}

// Numbers the new nodes of a level (first_node + rank among the level's nodes), tells every range's parent
// what its child is called, and queues the two halves of every node for the next level.
__global__ void k_emit(const Range* __restrict__ ranges, uint32_t n_ranges, const uint32_t* __restrict__ flags,
                       const uint32_t* __restrict__ ranks, const uint32_t* __restrict__ cuts, uint32_t first_node,
                       uint32_t* __restrict__ children, uint32_t* __restrict__ root_ref, Range* __restrict__ next) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_ranges) return;
  const Range r = ranges[i];
  uint32_t ref;
  if (flags[i]) {
    const uint32_t rank = ranks[i];
    ref = first_node + rank;
    next[2u * rank] = Range{r.a, cuts[i], ref, 0u};
    next[2u * rank + 1u] = Range{cuts[i], r.b, ref, 1u};
  } else {
    ref = leaf_ref(r.a, r.b);
  }
  if (r.parent == NO_PARENT) *root_ref = ref;
  else children[2u * (size_t)r.parent + r.side] = ref;
}

__device__ __forceinline__ DBox child_box(uint32_t ref, const BuildPrim* __restrict__ prims,
                                          const uint32_t* __restrict__ order, const DBox* __restrict__ node_boxes) {
  if (!(ref & 0x80000000u)) return node_boxes[ref];
  const uint32_t first = ref & 0x0FFFFFFFu, last = first + ((ref >> 28) & 7u);
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint32_t k = first; k <= last; ++k) {
    const BuildPrim p = prims[order[k]];
    for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], p.mn[a]); mx[a] = fmaxf(mx[a], p.mx[a]); }
  }
  return DBox{make_float4(mn[0], mn[1], mn[2], 0.0f), make_float4(mx[0], mx[1], mx[2], 0.0f)};
}

// Boxes of one level's nodes, deeper levels first: a node's children are leaves or nodes of deeper levels.
// Writes the node in the layout of scene_device.h.
__global__ void k_boxes(uint32_t first_node, uint32_t n_level_nodes, const uint32_t* __restrict__ children,
                        const BuildPrim* __restrict__ prims, const uint32_t* __restrict__ order,
                        DBox* __restrict__ node_boxes, float4* __restrict__ nodes) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_level_nodes) return;
  const uint32_t node = first_node + i;
  const uint32_t left = children[2u * (size_t)node], right = children[2u * (size_t)node + 1u];
  const DBox l = child_box(left, prims, order, node_boxes), r = child_box(right, prims, order, node_boxes);
  float4* out = nodes + 4 * (size_t)node;
  out[0] = make_float4(l.mn.x, l.mn.y, l.mn.z, __uint_as_float(left));
  out[1] = make_float4(l.mx.x, l.mx.y, l.mx.z, __uint_as_float(right));
  out[2] = make_float4(r.mn.x, r.mn.y, r.mn.z, 0.0f);
  out[3] = make_float4(r.mx.x, r.mx.y, r.mx.z, 0.0f);
  node_boxes[node] = DBox{make_float4(fminf(l.mn.x, r.mn.x), fminf(l.mn.y, r.mn.y), fminf(l.mn.z, r.mn.z), 0.0f),
                          make_float4(fmaxf(l.mx.x, r.mx.x), fmaxf(l.mx.y, r.mx.y), fmaxf(l.mx.z, r.mx.z), 0.0f)};
}

struct Scratch {  // frees what it holds on every way out
  void* p[16] = {};
  int n = 0;
  hipError_t alloc(void** out, size_t bytes) {
    const hipError_t e = hipMalloc(out, bytes ? bytes : 16);
    if (e == hipSuccess) p[n++] = *out;
    return e;
  }
  ~Scratch() { for (int k = 0; k < n; ++k) (void)hipFree(p[k]); }
};

#define GPU_TRY(call)                        \
  do {                                       \
    const hipError_t e_ = (call);            \
    if (e_ != hipSuccess) return e_;         \
  } while (0)

}  // namespace

hipError_t build_bvh_gpu(const std::vector<BuildPrim>& all_prims, uint32_t leaf_prims, uint32_t max_height,
                         hipStream_t stream, GpuBvh& out) {
  out = GpuBvh{};
  if (all_prims.empty()) return hipSuccess;
  leaf_prims = std::max(1u, std::min(leaf_prims, 8u));

  // the large ones: longest side above a quarter of the scene's, at most 64 of them
  std::vector<uint32_t> large;
  std::vector<BuildPrim> small_prims;
  std::vector<uint32_t> small_index;
  {
    float smn[3] = {INFINITY, INFINITY, INFINITY}, smx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const BuildPrim& p : all_prims)
      for (int a = 0; a < 3; ++a) { smn[a] = std::min(smn[a], p.mn[a]); smx[a] = std::max(smx[a], p.mx[a]); }
    // (the scene's box includes the large ones, so compare with what the rest would span: use the median-free
    // shortcut of the centre spread, which a single huge primitive hardly moves)
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const BuildPrim& p : all_prims)
      for (int a = 0; a < 3; ++a) {
        const float c = 0.5f * (p.mn[a] + p.mx[a]);
        clo[a] = std::min(clo[a], c);
        chi[a] = std::max(chi[a], c);
      }
    const float spread = std::max(std::max(chi[0] - clo[0], chi[1] - clo[1]), chi[2] - clo[2]);
    auto side = [](const BuildPrim& p) {
      return std::max(std::max(p.mx[0] - p.mn[0], p.mx[1] - p.mn[1]), p.mx[2] - p.mn[2]);
    };
    if (spread > 0.0f && all_prims.size() > 64)
      for (uint32_t i = 0; i < all_prims.size() && large.size() <= 64; ++i)
        if (side(all_prims[i]) > 0.25f * spread) large.push_back(i);
    if (large.size() > 64) large.clear();  // not a scene of a few big and many small things
    if (!large.empty()) {
      small_prims.reserve(all_prims.size() - large.size());
      small_index.reserve(all_prims.size() - large.size());
      size_t next_large = 0;
      for (uint32_t i = 0; i < all_prims.size(); ++i) {
        if (next_large < large.size() && large[next_large] == i) { ++next_large; continue; }
        small_prims.push_back(all_prims[i]);
        small_index.push_back(i);
      }
    }
  }
  const std::vector<BuildPrim>& prims = large.empty() ? all_prims : small_prims;
  const uint32_t n = (uint32_t)prims.size();

  // centroid bounds on the host: one pass over data that is in cache from the lowering
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (const BuildPrim& p : prims)
    for (int a = 0; a < 3; ++a) {
      const float c = 0.5f * (p.mn[a] + p.mx[a]);
      lo[a] = std::min(lo[a], c);
      hi[a] = std::max(hi[a], c);
    }
  // One cell size for the three axes (a cube over the longest side).  Scaling every axis to its own extent
  // would spend as many bits on a thin axis as on a long one: a field of marbles whose heights differ by a
  // hair would be cut by height at every third level, and the nodes would overlap in the other two.
  const float longest = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
  const float cells = longest > 0.0f && std::isfinite(longest) ? 2097151.0f / longest : 0.0f;
  const float scale[3] = {cells, cells, cells};

  Scratch scratch;
  BuildPrim* d_prims = nullptr;
  uint64_t *keys_a = nullptr, *keys_b = nullptr;
  uint32_t *vals_a = nullptr, *vals_b = nullptr;
  GPU_TRY(scratch.alloc((void**)&d_prims, (size_t)n * sizeof(BuildPrim)));
  GPU_TRY(scratch.alloc((void**)&keys_a, (size_t)n * 8));
  GPU_TRY(scratch.alloc((void**)&keys_b, (size_t)n * 8));
  GPU_TRY(scratch.alloc((void**)&vals_a, (size_t)n * 4));
  GPU_TRY(scratch.alloc((void**)&vals_b, (size_t)n * 4));
  GPU_TRY(hipMemcpyAsync(d_prims, prims.data(), (size_t)n * sizeof(BuildPrim), hipMemcpyHostToDevice, stream));
  const uint32_t grid_n = (n + kThreads - 1) / kThreads;
  hipLaunchKernelGGL(k_morton, dim3(grid_n), dim3(kThreads), 0, stream, d_prims, n, make_float3(lo[0], lo[1], lo[2]),
                     make_float3(scale[0], scale[1], scale[2]), keys_a, vals_a);
  GPU_TRY(hipGetLastError());

  hipcub::DoubleBuffer<uint64_t> keys(keys_a, keys_b);
  hipcub::DoubleBuffer<uint32_t> vals(vals_a, vals_b);
  size_t temp_bytes = 0;
  GPU_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys, vals, (int)n, 0, 63, stream));
  void* temp = nullptr;
  GPU_TRY(scratch.alloc(&temp, temp_bytes));
  GPU_TRY(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, vals, (int)n, 0, 63, stream));
  const uint32_t* order = vals.Current();

  const uint32_t n_top = (uint32_t)((large.size() + 7) / 8);  // nodes of the chain the large ones hang off
  if (max_height <= n_top + 1u) return hipErrorInvalidValue;
  const uint32_t tree_height_max = max_height - n_top;

  // ---- hierarchy, top down ----------------------------------------------------------------------------
  Range *ranges[2] = {nullptr, nullptr};
  uint32_t *flags = nullptr, *ranks = nullptr, *cuts = nullptr, *children = nullptr, *d_root = nullptr;
  DBox* node_boxes = nullptr;
  GPU_TRY(scratch.alloc((void**)&ranges[0], (size_t)n * sizeof(Range)));
  GPU_TRY(scratch.alloc((void**)&ranges[1], (size_t)n * sizeof(Range)));
  GPU_TRY(scratch.alloc((void**)&flags, (size_t)n * 4));
  GPU_TRY(scratch.alloc((void**)&ranks, (size_t)n * 4));
  GPU_TRY(scratch.alloc((void**)&cuts, (size_t)n * 4));
  GPU_TRY(scratch.alloc((void**)&children, (size_t)n * 8));
  GPU_TRY(scratch.alloc((void**)&d_root, 16));
  GPU_TRY(scratch.alloc((void**)&node_boxes, (size_t)n * sizeof(DBox)));
  size_t scan_bytes = 0;
  GPU_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, flags, ranks, (int)n, stream));
  void* scan_temp = nullptr;
  GPU_TRY(scratch.alloc(&scan_temp, scan_bytes));

  const Range whole{0u, n, NO_PARENT, 0u};
  GPU_TRY(hipMemcpyAsync(ranges[0], &whole, sizeof(whole), hipMemcpyHostToDevice, stream));
  std::vector<std::pair<uint32_t, uint32_t>> levels;  // (first node, nodes) per level
  uint32_t n_ranges = 1, n_nodes = 0;
  int cur = 0;
  while (n_ranges != 0) {
    const uint32_t level = (uint32_t)levels.size();
    if (level > tree_height_max) return hipErrorUnknown;  // cannot happen: the budget forces halving cuts
    const uint32_t grid = (n_ranges + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k_split, dim3(grid), dim3(kThreads), 0, stream, ranges[cur], n_ranges, keys.Current(), leaf_prims,
                       tree_height_max - level, flags, cuts);
    GPU_TRY(hipGetLastError());
    GPU_TRY(hipcub::DeviceScan::ExclusiveSum(scan_temp, scan_bytes, flags, ranks, (int)n_ranges, stream));
    hipLaunchKernelGGL(k_emit, dim3(grid), dim3(kThreads), 0, stream, ranges[cur], n_ranges, flags, ranks, cuts, n_nodes,
                       children, d_root, ranges[cur ^ 1]);
    GPU_TRY(hipGetLastError());
    uint32_t last_flag = 0, last_rank = 0;
    GPU_TRY(hipMemcpyAsync(&last_flag, flags + (n_ranges - 1), 4, hipMemcpyDeviceToHost, stream));
    GPU_TRY(hipMemcpyAsync(&last_rank, ranks + (n_ranges - 1), 4, hipMemcpyDeviceToHost, stream));
    GPU_TRY(hipStreamSynchronize(stream));
    const uint32_t level_nodes = last_rank + last_flag;
    if (level_nodes == 0) break;
    levels.emplace_back(n_nodes, level_nodes);
    n_nodes += level_nodes;
    n_ranges = 2 * level_nodes;
    cur ^= 1;
  }
  uint32_t height = (uint32_t)levels.size();

  float4* d_nodes = nullptr;
  GPU_TRY(hipMalloc((void**)&d_nodes, std::max<size_t>(16, (size_t)(n_nodes + n_top) * 64)));
  struct NodeGuard { float4* p; ~NodeGuard() { if (p) (void)hipFree(p); } } node_guard{d_nodes};

  // ---- boxes, bottom up ----------------------------------------------------------------------------------
  for (size_t l = levels.size(); l-- > 0;) {
    hipLaunchKernelGGL(k_boxes, dim3((levels[l].second + kThreads - 1) / kThreads), dim3(kThreads), 0, stream,
                       levels[l].first, levels[l].second, children, d_prims, order, node_boxes, d_nodes);
    GPU_TRY(hipGetLastError());
  }
  uint32_t root = 0;
  DBox root_box{};
  GPU_TRY(hipMemcpyAsync(&root, d_root, 4, hipMemcpyDeviceToHost, stream));
  out.order.resize(n);
  GPU_TRY(hipMemcpyAsync(out.order.data(), order, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
  if (n_nodes != 0) {
    GPU_TRY(hipMemcpyAsync(&root_box, node_boxes, sizeof(DBox), hipMemcpyDeviceToHost, stream));  // node 0 is the root
  } else {
    float bmn[3] = {INFINITY, INFINITY, INFINITY}, bmx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const BuildPrim& p : prims)
      for (int a = 0; a < 3; ++a) { bmn[a] = std::min(bmn[a], p.mn[a]); bmx[a] = std::max(bmx[a], p.mx[a]); }
    root_box = DBox{make_float4(bmn[0], bmn[1], bmn[2], 0.0f), make_float4(bmx[0], bmx[1], bmx[2], 0.0f)};
  }
  GPU_TRY(hipStreamSynchronize(stream));

  float mn[3] = {root_box.mn.x, root_box.mn.y, root_box.mn.z}, mx[3] = {root_box.mx.x, root_box.mx.y, root_box.mx.z};
  if (!large.empty()) {
    // tree order: the small ones as sorted, then the large ones; node k of the chain: left = a leaf of up to
    // eight large primitives, right = everything below
    for (uint32_t& o : out.order) o = small_index[o];
    std::vector<HostNode> top(n_top);
    for (uint32_t k = 0; k < n_top; ++k) {
      const uint32_t first = k * 8, last = std::min<uint32_t>(first + 8, (uint32_t)large.size());
      HostNode& nd = top[k];
      std::memset(&nd, 0, sizeof(nd));
      float lmn[3] = {INFINITY, INFINITY, INFINITY}, lmx[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (uint32_t j = first; j < last; ++j) {
        const BuildPrim& p = all_prims[large[j]];
        for (int a = 0; a < 3; ++a) { lmn[a] = std::min(lmn[a], p.mn[a]); lmx[a] = std::max(lmx[a], p.mx[a]); }
        out.order.push_back(large[j]);
      }
      // The leaf's box is given as the union with everything below.  Any superset is a valid box; this one makes
      // the walk, which takes the nearer child first and the left one on a tie, test the large primitives before
      // it goes down: they are the likeliest to end the ray early, and everything behind that hit is then culled.
      for (int a = 0; a < 3; ++a) {
        nd.lmin[a] = std::min(lmn[a], mn[a]); nd.lmax[a] = std::max(lmx[a], mx[a]);
        nd.rmin[a] = mn[a]; nd.rmax[a] = mx[a];
      }
      nd.left = 0x80000000u | ((last - first - 1u) << 28) | (n + first);
      nd.right = root;
      root = n_nodes + k;
      ++height;
      for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], lmn[a]); mx[a] = std::max(mx[a], lmx[a]); }
    }
    GPU_TRY(hipMemcpy(d_nodes + 4 * (size_t)n_nodes, top.data(), top.size() * sizeof(HostNode), hipMemcpyHostToDevice));
  }
  out.nodes = d_nodes;
  node_guard.p = nullptr;
  out.n_nodes = n_nodes + n_top;
  out.root_ref = root;
  out.height = height;
  for (int a = 0; a < 3; ++a) { out.mn[a] = mn[a]; out.mx[a] = mx[a]; }
  return hipSuccess;
}

}  // namespace mirt
