// Device-side view of an uploaded scene and of a render in flight.
// Shared by the HIP kernels (kernels.hip) and the host code that fills it
// (mirt_api.cpp).  Everything here is plain data passed by value as a kernel
// argument; pointers are device pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mirt.h"

namespace mirt {

// ---- child / hit references ---------------------------------------------------
// bit 31 set: leaf.  bits 28..30: tag.  bits 0..27: index inside the tag's array.
// bit 31 clear: index of an inner node.
constexpr uint32_t REF_LEAF = 0x80000000u;
constexpr uint32_t REF_NONE = 0xFFFFFFFFu;  // miss / empty tree
constexpr uint32_t TAG_SPHERE = 0, TAG_MOVING_SPHERE = 1, TAG_RECT = 2, TAG_BOX = 3, TAG_MEDIUM = 6;
constexpr uint32_t N_PRIM_TAGS = 4;
constexpr uint32_t MAX_PRIMS_PER_TAG = 1u << 28;

__host__ __device__ inline uint32_t make_leaf(uint32_t tag, uint32_t index) { return REF_LEAF | (tag << 28) | index; }
__host__ __device__ inline uint32_t ref_tag(uint32_t ref) { return (ref >> 28) & 7u; }
__host__ __device__ inline uint32_t ref_index(uint32_t ref) { return ref & 0x0FFFFFFFu; }

// material word: bits 28..31 material kind, bits 0..27 material index
__host__ __device__ inline uint32_t make_material_word(uint32_t kind, uint32_t index) { return (kind << 28) | index; }

// Traversal stack entries per lane (kept in LDS).  The host builder never
// produces a tree deeper than this.
constexpr int STACK_DEPTH = 32;
constexpr int BLOCK_THREADS = 256;

// BVH node, 64 bytes = 4 x float4:
//   n0 = left.min.xyz,  bits(left ref)
//   n1 = left.max.xyz,  bits(right ref)
//   n2 = right.min.xyz, unused
//   n3 = right.max.xyz, unused
// Child boxes live in the parent so one 64-byte fetch decides both children.

// Per-primitive side data (one uint4 per primitive, per tag).
//   x = material word, y = chain index, z = order (tie-break rank), w = flags
// ids: uint2 (type_index, object_index) per primitive, per tag.

struct DMaterial {  // 16 B
  uint32_t kind;
  uint32_t texture;
  float param;
  uint32_t pad;
};

struct DTexture {  // 32 B
  uint32_t kind;
  uint32_t odd;
  uint32_t even;
  uint32_t index;
  float r, g, b;
  float scale;
};

struct DImage {  // 16 B
  uint64_t offset;  // byte offset into image_data
  uint32_t width;
  uint32_t height;
};

struct DLight {  // 32 B
  uint32_t kind;
  float p[5];
  uint32_t pad[2];
};

struct DMedium {
  uint32_t boundary_ref;  // leaf ref of the boundary primitive
  float neg_inv_density;
  uint32_t material_word;
  uint32_t chain;  // wrappers around the medium itself
  uint32_t order;
  uint32_t type_index;
  uint32_t object_index;
  uint32_t pad;
};

struct DeviceScene {
  const float4* nodes;
  uint32_t root_ref;
  uint32_t n_chains;  // 1 = only the empty chain: skip chain lookups entirely

  const float4* sphere;   // 1 float4 per sphere: center, radius
  const float4* msphere;  // 3 per moving sphere: (c0, radius) (c1, time0) (time1,-,-,-)
  const float4* rect;     // 2 per rect: (a0,a1,b0,b1) (k, bits(axis), -, -)
  const float4* box;      // 2 per box: (min,-) (max,-)
  // per tag (sphere, moving sphere, rect, box); separate members rather than
  // an array so a run-time tag selects a pointer without a private-memory copy
  const uint4* info_sphere;
  const uint4* info_msphere;
  const uint4* info_rect;
  const uint4* info_box;
  const uint2* ids_sphere;
  const uint2* ids_msphere;
  const uint2* ids_rect;
  const uint2* ids_box;

  const uint2* chains;      // (first_op, n_ops)
  const float4* chain_ops;  // (bits(kind), p0, p1, p2)

  uint32_t n_media;
  const DMedium* media;

  const DMaterial* materials;
  const DTexture* textures;
  const float4* perlin_vec;     // 256 per table (xyz, -)
  const uint32_t* perlin_perm;  // 768 per table: perm_x, perm_y, perm_z
  const DImage* images;
  const uint8_t* image_data;

  uint32_t lights_mode;
  uint32_t n_lights;
  uint32_t n_light_groups;
  const DLight* lights;
  const uint2* light_groups;  // (first_light, n_lights)
};

// Per-iteration counters, double buffered by iteration parity.
struct IterCounters {
  uint32_t n_cur;        // paths carried over from the previous iteration (filled by shade)
  uint32_t queue_n[4];   // entries in each material queue (filled by extend)
  uint32_t pad[3];
  uint64_t issued;       // camera paths started before this iteration
  uint64_t pad2;
};

struct WorkCounters {
  unsigned long long segments;
  unsigned long long nodes;
  unsigned long long prims;
  unsigned long long pad;
};

constexpr int QUEUE_LAMBERTIAN = 0, QUEUE_METAL = 1, QUEUE_DIELECTRIC = 2, QUEUE_TERMINATE = 3, N_QUEUES = 4;

struct RenderConst {
  mirt_camera cam;
  uint32_t width, height;
  uint32_t first_sample;
  uint32_t max_depth;
  float background[3];
  uint32_t key0, key1;
  uint32_t pool;             // path records per buffer
  uint64_t total_paths;      // width*height*samples_per_pixel of this call
};

// Path record, 64 bytes = 4 x float4:
//   r0 = origin.xyz, time
//   r1 = direction.xyz, t of the closest hit (written by extend)
//   r2 = throughput.rgb, bits(pixel = y*width + x, y up)
//   r3 = bits(sample index), bits(bounce), bits(hit ref), unused

// ---- launchers implemented in kernels.hip ------------------------------------
void launch_extend(const DeviceScene& sc, const RenderConst& rc, float4* cur, IterCounters* counters, int parity,
                   WorkCounters* work, uint32_t* queues, bool count_work, int grid, hipStream_t stream);
void launch_shade(const DeviceScene& sc, const RenderConst& rc, const float4* cur, float4* next,
                  IterCounters* counters, int parity, const uint32_t* queues, float* rgb_sum, int grid,
                  hipStream_t stream);
void launch_resolve(const float* rgb_sum, uint32_t n_values, uint32_t spp, uint8_t* rgb24, hipStream_t stream);
void launch_trace(const DeviceScene& sc, const mirt_ray* rays, size_t n, uint32_t key0, uint32_t key1,
                  uint32_t* type_index, uint32_t* object_index, float* t, hipStream_t stream);
int extend_blocks_per_cu();
int shade_blocks_per_cu();

}  // namespace mirt
