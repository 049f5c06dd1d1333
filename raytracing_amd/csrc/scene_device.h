// Device-side view of an uploaded scene and of a render in flight.
// Shared by the HIP kernels (kernels.hip) and the host code that fills it
// (mirt_api.cpp).  Everything here is plain data; pointers are device pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mirt.h"

namespace mirt {

// ---- references ------------------------------------------------------------------
// Inside the tree (node children, traversal stack):
//   bit 31 clear: index of an inner node
//   bit 31 set:   a leaf = (count - 1) in bits 28..30, index of its first primitive record in bits 0..27
//                 (a leaf's primitives are consecutive records)
// Hit references (what the walk hands to shading, what a trace resolves to ids):
//   bit 31 set, primitive kind (TAG_*) in bits 28..30, record index in bits 0..27;
//   TAG_MEDIUM + medium number for a constant medium; REF_NONE for a miss.
constexpr uint32_t REF_LEAF = 0x80000000u;
constexpr uint32_t REF_NONE = 0xFFFFFFFFu;  // miss / empty tree / traversal finished
constexpr uint32_t TAG_SPHERE = 0, TAG_MOVING_SPHERE = 1, TAG_RECT = 2, TAG_BOX = 3, TAG_MEDIUM = 6;
constexpr uint32_t N_PRIM_TAGS = 4;
constexpr uint32_t MAX_PRIMS = (1u << 28) - 1;
constexpr uint32_t LEAF_MAX_PRIMS = 4;  // primitives per leaf the host builder aims for (the encoding allows 8)

__host__ __device__ inline uint32_t make_leaf(uint32_t tag, uint32_t index) { return REF_LEAF | (tag << 28) | index; }
__host__ __device__ inline uint32_t ref_tag(uint32_t ref) { return (ref >> 28) & 7u; }
__host__ __device__ inline uint32_t ref_index(uint32_t ref) { return ref & 0x0FFFFFFFu; }

// material word: bits 28..31 the queue a hit on this material is shaded in (QUEUE_*), bits 0..27 material index
__host__ __device__ inline uint32_t make_material_word(uint32_t queue, uint32_t index) { return (queue << 28) | index; }

// Traversal stack entries per lane (kept in LDS).  The host builder never
// produces a tree taller than STACK_DEPTH - 2.
constexpr int STACK_DEPTH = 32;
// Scenes walked from global memory have four-child nodes (WideNode, bvh_build.h): a step may leave three children
// waiting, so the stack holds up to 3 * height + 1 entries.  The first WIDE_LDS_STACK of them are kept in LDS (1024
// lanes x 4 bytes each), deeper ones in global memory.  A walk put aside carries up to SAVE_STACK entries.
constexpr int WIDE_LDS_STACK = 38;
constexpr int WIDE_MAX_STACK = 3 * (STACK_DEPTH - 2) + 1;
constexpr uint32_t SAVE_STACK = 40;
constexpr int RENDER_THREADS = 1024;    // render: one block per CU, a CU's worth of waves (the most a block may have)
constexpr int RENDER_BLOCKS = 256;
constexpr size_t LDS_BYTES_PER_CU = 160 * 1024;

// BVH node, 64 bytes = 4 x float4:
//   n0 = left.min.xyz,  bits(left ref)
//   n1 = left.max.xyz,  bits(right ref)
//   n2 = right.min.xyz, growth of the left box
//   n3 = right.max.xyz, growth of the right box
// Child boxes live in the parent so one 64-byte fetch decides both children.
//
// The box test must never reject a ray whose primitive test, in the reference's f32 arithmetic, reports a hit:
// the reference itself never culls (aabb.rs:41-51 is always true on this path).  So every box is grown, per ray,
// by  L * (L * growth + BOX_GROW_LINEAR),  L = the largest |coordinate| of the box relative to the ray origin:
//  * Sphere test (objects.rs:42-56): with os = origin - centre, M = max(|os|, r) and u = 2^-24, the computed
//    discriminant half_b^2 - a*c is off by e from |d|^2 (r^2 - m^2), m the distance of the line from the centre:
//    half_b is off by 3 u |os| |d| (three roundings on the longest path of the dot product), so half_b^2 by
//    6 u + 1 u; a by 3 u |d|^2 and c = |os|^2 - r^2 by 3 u + 1 u + 1 u, so a*c by 3 u + 5 u + 1 u; the subtraction
//    adds 1 u: |e| <= 17 u M^2 |d|^2 (and terms in u^2).  Take |d| = 1.  The root the test reports is off by
//    e_t <= 17 u M along the ray (half_b, the square root, the subtraction, a, the division), so the point P it
//    stands for has  |P - centre|^2 = m^2 + (sqrt(disc) + e_t)^2 = r^2 + e + 2 h e_t + e_t^2 + 2 u h^2,
//    h <= sqrt(r^2 + e) the computed half-chord: with A = 17.1 u M^2 and B = 36 u M,
//    |P - centre| - r <= sqrt(r^2 + A + B r) - r.
//      - Any radius: that is at most sqrt(A) + B / 2 = 1.0107e-3 M; the rounding of origin - centre (u M) added,
//        P lies within 1.011e-3 M of the ball.
//      - A given radius: it is also at most (A + B r) / (2 r), so P lies within 8.6 u M^2 / r + 19 u M.
//    Scenes staged in LDS use the second: M^2 <= 3 L^2 makes it 25.7 u L^2 / r + 33 u L, and growth =
//    BOX_GROW_SPHERE / r = 64 u / r leaves a factor 2.5 on the quadratic part, with r the smallest sphere radius
//    below the box (a larger box has a larger L and a smaller r: its growth covers every sphere in it).
//  * Rects and box sides (objects.rs:206-228): t = (k - o) / d and the point o + t d are off by at most
//    4 u (|o| + L) <= 8 u L + 4 u |box|; wrapped primitives see a ray whose rotation and translation rounded by
//    6 u |o|; the box test's own subtractions, reciprocals and fused multiply-adds are off by 4 u L.
//    BOX_GROW_LINEAR = 64 u covers the parts in L (spheres 33 u + 4 u, rects 8 u + 4 u), the builder's own padding
//    of 170 u |box| (Bounds::pad) the rest.
// Scenes walked from global memory use |P - centre| - r <= sqrt(r^2 + A + B r) - r itself, with F for M -- the
// Euclidean length of the farthest |coordinates| of the node's boxes relative to the ray origin (F >= M: the box
// holds the ball) -- and the smallest sphere radius of the scene for r (the expression falls as r grows, since
// B^2 < 4 A; a scene without spheres has r huge and the term all but vanishes):
//   pad = HALO_MARGIN * (sqrt(r^2 + x) - r) + BOX_GROW_FAR_LINEAR * F,   x = F (HALO_QUADRATIC F + HALO_LINEAR r),
// HALO_QUADRATIC = 17.1 u, HALO_LINEAR = 36 u, 5 % on top as the LDS rule's factor stands for, and a linear part
// for the rounding of origin - centre (u F), for rects and box sides (64 u L) and for the box test's own arithmetic
// (4 u F and the decoded planes' rounding, which has a term of its own: BOX_DECODE_SLACK).  For r -> 0 this is the
// "any radius" bound, 1.06e-3 F; a thousand units from a sphere of radius 0.2 it is 0.83 F / 1000 + ..., four
// hundred units away 0.25 where the radius-free bound has 0.40: the f32 test no longer resolves r^2 in |os|^2 - r^2
// there, and these are the hits it may really report.
constexpr float BOX_GROW_SPHERE = 3.814697265625e-06f;  // 2^-18
constexpr float BOX_GROW_LINEAR = 3.814697265625e-06f;  // 2^-18
constexpr float HALO_QUADRATIC = 1.0192394e-06f;         // 17.1 * 2^-24
constexpr float HALO_LINEAR = 2.1457672e-06f;            // 36 * 2^-24
constexpr float HALO_MARGIN = 1.05f;
constexpr float BOX_GROW_FAR_LINEAR = 4.4703484e-06f;    // 75 * 2^-24
// Four-child nodes are tested relative to the ray origin, a plane in one fused multiply-add (wide_step): what that
// adds to the rounding of a plane, per unit of |corner - origin| + |corner| + 255 * grid spacing: 3 u = 3 * 2^-24 and
// 5 % (derivation at wide_step).
constexpr float BOX_DECODE_SLACK = 1.878e-7f;

// Primitive records: 4 x float4 = one 64-byte line per primitive, in the order the tree lists them
// (so a leaf's primitives are neighbours); boundary-only primitives follow at the end.
//   qa = sphere (center, radius) | moving sphere (c0, radius) | rect (a0,a1,b0,b1) | box (min, -)
//   qb =                         | moving sphere (c1, time0)  | rect (k, bits(axis),-,-) | box (max, -)
//   qc = bits(chain << 2 | kind), bits(order = tie-break rank), bits(material word), moving sphere: time1
//   qd = bits(flags), -, -, -
// Traversal and shading need qa..qc only; when the scene fits, the render kernel keeps those 48 bytes per
// primitive in LDS.
// prim_ids: uint2 (type_index, object_index) per record (only read when ids are reported).
constexpr uint32_t PRIM_QUADS = 4;      // record stride in global memory
constexpr uint32_t PRIM_QUADS_LDS = 3;  // record stride of the LDS copy

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

  uint32_t n_nodes;     // nodes of one copy of the tree
  uint32_t bvh_height;  // inner nodes on the longest root-to-leaf path
  // Scenes with moving spheres keep n_slices copies of the node array, copy s fitted to where the spheres are
  // during slice s of [slice_time0, slice_time0 + n_slices / slice_scale]: a ray walks the copy of its time.
  uint32_t n_slices;
  float slice_time0, slice_scale;
  uint32_t wide;  // nodes are four-child WideNodes (the scene is walked from global memory)
  uint32_t n_perlins;  // noise tables (256 vectors and 3 x 256 permutation entries each)
  float min_radius;   // smallest radius of a sphere in the tree (at most 1e15; that much for a scene without spheres)
  uint32_t keep_out;  // never staged in LDS, whatever its size (MIRT_NO_LDS at upload: a small scene through the large scenes' code)

  const float4* prims;  // primitive records, layout above
  const uint2* prim_ids;
  uint32_t n_prims;     // records a leaf can point at (boundary-only records come after them)
  uint32_t n_chain_ops;

  const float4* chains;     // two quads per chain: (offset.xyz, bits(shape | n_ops << 8 | first_op << 12)), (sin, cos, -, -)
  const float4* chain_ops;  // (bits(kind), p0, p1, p2)

  uint32_t n_media;
  const DMedium* media;

  uint32_t n_materials, n_textures;
  const DMaterial* materials;
  const DTexture* textures;
  const float4* perlin_vec;     // 256 per table (xyz, -)
  const uint32_t* perlin_perm;  // 768 per table: perm_x, perm_y, perm_z
  const DImage* images;
  const uint8_t* image_data;

  float root_lo[3], root_hi[3];  // bounds of everything in the tree

  uint32_t lights_mode;
  uint32_t n_lights;
  uint32_t n_light_groups;
  const DLight* lights;
  const uint2* light_groups;  // (first_light, n_lights)
};

// ---- the path pool ----------------------------------------------------------------
// The pool is split into equal regions of consecutive slots, one per wave of the render kernel (RenderConfig::regions).
// A region is private to its wave: two buffers of path records (this round's and the next one's) and N_QUEUES
// shading queues of path records.  The only thing the waves share is the counter new paths are numbered from.
constexpr uint32_t N_REGIONS = RENDER_BLOCKS * (RENDER_THREADS / 64);  // of a launch with full blocks

struct RenderState {
  unsigned long long issued;  // camera paths handed out so far (may run past the render's total)
  unsigned long long pad[15];
};

struct WorkCounters {
  unsigned long long segments;
  unsigned long long nodes;
  unsigned long long prims;
  unsigned int rounds;  // most start/walk/shade rounds any block went through
  unsigned int pad;
  unsigned long long wave_nodes;  // wave-wide turns spent on tree steps / on primitive tests
  unsigned long long wave_prims;
  // shader-clock cycles summed over waves, per phase of the render kernel (instrumented build only):
  // 0 start paths + fetch rays, 1 tree steps, 2 primitive tests, 3 shade, 4 hand-in + binning, 5 whole kernel
  unsigned long long cycles[MIRT_PHASE_COUNT];
  // diagnostic build (MIRT_CHECKED) only: site of the first index found out of range (0 = none), the three values
  // recorded there, block and thread that found it, number of violations in all
  unsigned long long check[8];
  // diagnostic build (MIRT_SURVIVOR_AUDIT) only, per wave of a block: survivors shading counted, slots among them
  // that no lane wrote
  unsigned long long audit[32];
};

// Hits are shaded queue by queue so that a wave runs one material's code: Lambertian surfaces of one colour,
// metal, glass, what ends the path (lights, Isotropic, misses), Lambertian surfaces with a texture to evaluate;
// single-colour Lambertian hits are split once more by the primitive hit (spheres and media / rects and boxes: the
// hit records are worked out by different code, and the survivors of one queue start near each other, which keeps
// the next walk's waves on the same kind of primitive).  The material word names one of the material queues.
// QUEUE_CONTINUE holds rays of a large scene whose walk used up its step budget this round: they go back
// into the pool unshaded and walk their next stretch in the next round.  QUEUE_MISS: rays that hit nothing (only
// the media and the background are left for them; apart from the surfaces that end a path in QUEUE_TERMINATE).
// QUEUE_CONTINUE comes first: its rays move into the first slots of the next round's buffer, so the next walk starts
// with them -- the rays with the longest walks still ahead of them -- and the short ones fill the round's end.
constexpr int QUEUE_CONTINUE = 0, QUEUE_LAMBERTIAN = 1, QUEUE_METAL = 2, QUEUE_DIELECTRIC = 3, QUEUE_TERMINATE = 4,
              QUEUE_LAMBERTIAN_TEXTURED = 5, QUEUE_LAMBERTIAN_FLAT = 6, QUEUE_MISS = 7, N_QUEUES = 8;

// Division of a 32-bit number by a divisor fixed for the whole launch: a multiply-high and two shifts in
// place of the long sequence a division compiles to (Granlund & Montgomery, unsigned case).
struct FastDiv {
  uint32_t m, sh1, sh2;
};
inline FastDiv make_fastdiv(uint32_t d) {  // d >= 1
  uint32_t l = 0;
  while (l < 32 && (1ull << l) < d) ++l;
  const uint64_t m = ((((1ull << l) - d) << 32) / d) + 1;
  return FastDiv{(uint32_t)m, l < 1 ? l : 1u, l > 0 ? l - 1 : 0u};
}

struct RenderConst {
  mirt_camera cam;
  uint32_t width, height;
  uint32_t band_y0, band_rows;  // the rows this call renders: y in [band_y0, band_y0 + band_rows), y = 0 at the bottom
  uint32_t first_sample;
  uint32_t max_depth;
  float background[3];
  uint32_t key0, key1;
  uint32_t pool;             // path records per buffer = N_REGIONS * region_cap
  uint32_t region_cap;       // slots per region; region s owns slots [s * region_cap, (s + 1) * region_cap)
  uint64_t total_paths;      // width*band_rows*samples_per_pixel of this call
  // Order in which a pass walks the image: group j of 64 pixels (of the n_groups whole groups) is visited
  // as number (j * group_stride) % n_groups; group_stride is coprime to n_groups and their product < 2^32.
  uint32_t n_groups, group_stride;
  FastDiv by_pixels, by_groups, by_band, by_width;  // / (width*band_rows), / n_groups, / (8*width), / width
  // opt-in estimator changes (MIRT_RENDER_EVEN_LIGHTS, MIRT_RENDER_DROP_NONFINITE) and the jitter strata per axis
  // (0 = the reference's unstratified jitter)
  uint32_t estimator, strata;
  FastDiv by_strata, by_cells;  // / strata, / strata^2
};

// Path record, 64 bytes = 4 x float4:
//   r0 = origin.xyz, time
//   r1 = direction.xyz, bits(pixel = y*width + x, y up)
//   r2 = throughput.rgb, bits(sample index << 8 | bounce)
//   r3 = t of the closest surface, bits(hit ref), bits(side of a box hit), -     (only in the copy the walk puts
//        into a shading queue; the two record buffers carry r0..r2)
// Shading queue q of region s holds whole path records (the walk moves a finished ray's record there, hit
// included) at queues[4 * (s * N_QUEUES + q) * region_cap ...]: shading reads a queue front to back.
// A walk put aside: (reference in hand, stack entries in use, best ref, best order), (best t, best material word,
// best box side, -), then up to SAVE_STACK stack entries, four to a quad.
constexpr uint32_t SAVE_QUADS = 2 + SAVE_STACK / 4;
constexpr uint32_t MAX_SAMPLE_INDEX = (1u << 24) - 1u;
constexpr uint32_t MAX_DEPTH_LIMIT = 255;

// ---- the fixed workspace ----------------------------------------------------------
// For callers that may not allocate device memory: a file-scope device array in kernels.hip, one copy per GPU,
// holding a path pool of STATIC_POOL_PATHS records (two record buffers and N_QUEUES queues: 640 bytes a path)
// behind a control page with the render's counters and launch arguments.
// The render kernel runs there with four waves to a block, each with a region of 64 paths.
constexpr uint32_t STATIC_POOL_PATHS = 65536;
constexpr uint32_t STATIC_WAVES_PER_BLOCK = 4;
constexpr uint32_t STATIC_REGIONS = RENDER_BLOCKS * STATIC_WAVES_PER_BLOCK;
constexpr uint32_t STATIC_SAVE_CAP = 16;  // walks put aside per region and round in the fixed workspace
constexpr size_t STATIC_CONTROL_BYTES = 4096;
constexpr size_t STATIC_SAVE_BYTES = (size_t)STATIC_REGIONS * 2 * STATIC_SAVE_CAP * 16 * (2 + 40 / 4);
constexpr size_t STATIC_WORKSPACE_BYTES =
    STATIC_CONTROL_BYTES + (size_t)STATIC_POOL_PATHS * 64 * (2 + N_QUEUES) + STATIC_SAVE_BYTES;
// Address of the current device's copy (`device` must be the current device); looked up once per device.
hipError_t static_workspace_address(int device, void** out);

// ---- launchers implemented in kernels.hip ------------------------------------
// How the render kernel is launched for a scene: one block per CU, one region per wave, with nodes, primitive
// records, wrapper chains, materials, textures and the light list staged in LDS when they fit beside the block's
// traversal stacks, else read from global memory.
constexpr uint32_t PERLIN_LDS_BYTES = 256 * 16 + 768;  // one noise table in LDS
struct RenderConfig {
  bool lds_scene;
  int block;  // threads per block: a multiple of 64, at most RENDER_THREADS
  int grid;
  uint32_t regions() const { return (uint32_t)grid * ((uint32_t)block / 64u); }
  int stack_depth;
  size_t lds_bytes;
  uint32_t refill_min;  // a wave hands in finished rays and fetches new ones once this many lanes have none
  uint32_t node_min;    // a wave keeps stepping down the tree while this many of its lanes hold an inner node
  uint32_t prims_per_turn;  // primitive tests a lane makes, one after the other, between two runs of tree steps
  // Scenes staged in LDS with room to spare: the noise tables go behind the stacks, vectors as they are, the
  // permutations as bytes (4864 bytes a table); byte offset of the first table in the block's LDS, 0 = not staged
  uint32_t lds_perlin_offset;
};
RenderConfig plan_render(const DeviceScene& sc);
// Everything the render kernel is told, read by it from device memory.
struct RenderArgs {
  DeviceScene sc;
  RenderConst rc;
  float4* paths_a;   // the two buffers of path records, pool * 64 bytes each
  float4* paths_b;
  float4* queues;    // pool * N_QUEUES path records with their hits, 64 bytes each
  RenderState* state;
  WorkCounters* work;
  float* rgb_sum;
  uint32_t refill_min, node_min, prims_per_turn;
  uint32_t lds_perlin_offset;  // RenderConfig::lds_perlin_offset
  uint32_t split_survivors;    // shading keeps the survivors that head down apart from those that head up (WaveShared)
  uint32_t waves_per_block;  // block size of the launch / 64; wave w of block b owns region b * waves_per_block + w
  // Scenes walked from global memory: a ray takes at most this many steps (tree steps + primitive tests) per
  // round; then its walk is put aside in `saves` (save_cap entries of SAVE_QUADS quads per region and round
  // parity) and goes on in the next round, so that the few rays that graze a large scene for thousands of steps do
  // not hold up the round for everyone; 0 = every ray is walked to the end.
  uint32_t step_budget;
  uint32_t save_cap;
  uint4* saves;
  // stack levels beyond WIDE_LDS_STACK: overflow_levels rows of RENDER_THREADS entries per block, or null when the
  // tree is flat enough never to need them
  uint32_t* stack_overflow;
  uint32_t overflow_levels;
  // Tracing caller-supplied rays instead of rendering (rays != nullptr): rc.total_paths rays come from `rays`,
  // each is walked once, and what ray g hit goes to hit_type[g], hit_object[g], hit_t[g] (and hit_records).
  const mirt_ray* rays;
  uint32_t* hit_type;
  uint32_t* hit_object;
  float* hit_t;
  float4* hit_records;  // optional: four quads per ray, the layout of mirt_hit_record
};
// `args` is a device pointer; the caller keeps it unchanged until the kernel has finished.
hipError_t launch_render(const RenderArgs* args, bool count_work, const RenderConfig& cfg, hipStream_t stream);
void launch_resolve(const float* rgb_sum, uint32_t n_values, uint32_t spp, uint8_t* rgb24, hipStream_t stream);
void launch_add(float* dst, const float* src, size_t n_values, hipStream_t stream);  // dst += src


}  // namespace mirt
